// Window triggerers and the open-window record.
// Semantics: SURVEY.md A.2/A.3, reference /root/reference/wf/window.hpp:48-298.
//   count-based: id <  base           -> OLD   (impossible on in-order streams)
//                id <= base + win - 1 -> IN, else FIRED
//   time-based : ts <  base -> OLD; ts < base+win -> IN;
//                ts <  base+win+delay -> DELAYED; else FIRED
// with base = initial_id + lwid * slide.
#pragma once
#include <cstdint>

#include "basic.hpp"

namespace wf {

struct Triggerer_CB {
    uint64_t win_len, slide_len, lwid, initial_id;
    Triggerer_CB(uint64_t w, uint64_t s, uint64_t l, uint64_t init) : win_len(w), slide_len(s), lwid(l), initial_id(init) {}
    win_event_t operator()(uint64_t id) const {
        const uint64_t base = initial_id + lwid * slide_len;
        if (id < base) return OLD;
        if (id <= base + win_len - 1) return IN;
        return FIRED;
    }
};

struct Triggerer_TB {
    uint64_t win_len, slide_len, lwid, initial_ts, triggering_delay;
    Triggerer_TB(uint64_t w, uint64_t s, uint64_t l, uint64_t init, uint64_t delay)
        : win_len(w), slide_len(s), lwid(l), initial_ts(init), triggering_delay(delay) {}
    win_event_t operator()(uint64_t ts) const {
        const uint64_t base = initial_ts + lwid * slide_len;
        if (ts < base) return OLD;
        if (ts < base + win_len) return IN;
        if (ts < base + win_len + triggering_delay) return DELAYED;
        return FIRED;
    }
};

// One open window of one key.  The result is pre-stamped with (key, gwid, ts);
// for time-based windows ts is the last instant of the window, for count-based
// ones it follows the last tuple that fell IN (reference window.hpp:162-167, :202).
template <typename result_t, typename key_t>
struct Window {
    uint64_t lwid = 0;
    uint64_t gwid = 0;
    uint64_t start = 0;  // first ordinal (id/ts) of the window
    uint64_t end = 0;    // one past the last ordinal
    size_t no_tuples = 0;
    bool batched = false;  // GPU engines: already scheduled in a device batch
    result_t result{};

    Window(const key_t& key, uint64_t lwid_, uint64_t gwid_, uint64_t start_, uint64_t win_len, win_type_t type)
        : lwid(lwid_), gwid(gwid_), start(start_), end(start_ + win_len) {
        result.setControlFields(key, gwid_, type == CB ? 0 : end - 1);
    }
    win_event_t onOrdinal(uint64_t ord, win_type_t type, uint64_t delay) const {
        if (batched) return BATCHED;
        // the window carries its own bounds: a triggerer anchored at `start` with lwid 0
        if (type == CB) return Triggerer_CB(end - start, 0, 0, start)(ord);
        return Triggerer_TB(end - start, 0, 0, start, delay)(ord);
    }
};

}  // namespace wf
