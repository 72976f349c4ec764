// Win_Seq: the sequential window engine (one replica = one thread) and the
// stand-alone Win_Seq operator.
//
// Semantics follow /root/reference/wf/win_seq.hpp:304-529 and SURVEY.md A.2-A.7:
// window assignment under (possibly nested) partitioning through
// WinOperatorConfig, late-tuple dropping, sliding/tumbling vs hopping windows,
// non-incremental functions over an Iterable of the archived tuples or
// incremental per-tuple updates, result renumbering for the PLQ and MAP roles,
// end-of-stream markers that only advance window state, flush of every open
// window at end-of-stream.  The implementation is ours: windows carry their
// ordinal bounds, so the archive is addressed with binary searches on [start,end)
// instead of tracking first/last tuples per window, and one code path serves
// count- and time-based windows.
#pragma once
#include <cmath>
#include <deque>
#include <unordered_map>

#include "basic_operator.hpp"
#include "iterable.hpp"
#include "meta.hpp"
#include "simple_operator.hpp"
#include "stream_archive.hpp"
#include "window.hpp"

namespace wf {

// parameters shared by every window engine
struct WinParams {
    uint64_t win_len = 1;
    uint64_t slide_len = 1;
    uint64_t triggering_delay = 0;
    win_type_t winType = CB;
    WinOperatorConfig config;  // default: a single replica seeing the whole stream
    role_t role = SEQ;
    // MAP role: first id emitted and stride between ids (reference setMapIndexes)
    uint64_t map_first = 0, map_stride = 1;
};

// window-assignment arithmetic of SURVEY.md A.4 (reference win_seq.hpp:328-336)
struct WinAssign {
    uint64_t first_gwid_key;
    uint64_t initial_id;
    uint64_t gwid_stride;
    WinAssign(const WinOperatorConfig& c, size_t hashcode, role_t role) {
        const uint64_t a_inner = (c.id_inner + c.n_inner - (hashcode % c.n_inner)) % c.n_inner;
        const uint64_t a_outer = (c.id_outer + c.n_outer - (hashcode % c.n_outer)) % c.n_outer;
        first_gwid_key = a_inner * c.n_outer + a_outer;
        const uint64_t initial_outer = a_outer * c.slide_outer;
        const uint64_t initial_inner = a_inner * c.slide_inner;
        initial_id = (role == WLQ || role == REDUCE) ? initial_inner : initial_outer + initial_inner;
        gwid_stride = (uint64_t)c.n_outer * c.n_inner;
    }
};

template <typename tuple_t, typename result_t, typename F>
class Win_Seq_Node : public Replica_Base {
public:
    static constexpr bool isNIC = meta::win_sig<F>::is_nic;
    using key_t = key_of_t<tuple_t>;

    Win_Seq_Node(F func, WinParams p, std::string name, size_t par, size_t idx, meta::closing_func_t closing, std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub), true), func_(std::move(func)), p_(p) {
        if (p_.win_len == 0 || p_.slide_len == 0) fatal_error("window length or slide cannot be zero");
        set_io_bytes(sizeof(tuple_t), sizeof(result_t));
    }

    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        const bool marker = is_marker(msg);
        const key_t key = key_of(*t);
        const size_t hashcode = std::hash<key_t>()(key);
        const uint64_t ord = (p_.winType == CB) ? id_of(*t) : ts_of(*t);
        auto it = keys_.find(key);
        if (it == keys_.end()) it = keys_.emplace(key, KeyD(p_)).first;
        KeyD& kd = it->second;
        const WinAssign wa(p_.config, hashcode, p_.role);
        // late tuple? (reference :338-345)
        const uint64_t min_boundary = (kd.last_lwid >= 0) ? p_.win_len + (uint64_t)kd.last_lwid * p_.slide_len : 0;
        if (ord < wa.initial_id + min_boundary) {
            if (kd.last_lwid >= 0 && !marker) dropped_++;
            tuple_delete(t);
            return;
        }
        // last window containing (or preceding) the tuple
        long last_w;
        const uint64_t rel = ord - wa.initial_id;
        bool in_gap = false;
        if (p_.win_len >= p_.slide_len) {
            last_w = (long)((rel + 1 + p_.slide_len - 1) / p_.slide_len) - 1;  // ceil((rel+1)/slide) - 1
        } else {
            // hopping windows: a tuple that falls in the gap after window n belongs to no window, but - like an
            // end-of-stream marker - it still proves that window n is over: the windows up to n are opened and fired
            // before it is discarded.  (The reference discards it first, win_seq.hpp:352-365, so its Key_Farm emits
            // fewer trailing empty windows than its Win_Farm on the same stream; here every operator agrees.)
            const uint64_t n = rel / p_.slide_len;
            last_w = (long)n;
            in_gap = rel >= n * p_.slide_len + p_.win_len;
        }
        if constexpr (isNIC) {
            if (!marker && !in_gap) kd.archive.insert(*t);
        }
        // open the windows this tuple reaches
        for (long lwid = (long)kd.next_lwid; lwid <= last_w; lwid++) {
            const uint64_t gwid = wa.first_gwid_key + (uint64_t)lwid * wa.gwid_stride;
            kd.wins.emplace_back(key, (uint64_t)lwid, gwid, wa.initial_id + (uint64_t)lwid * p_.slide_len, p_.win_len, p_.winType);
            kd.next_lwid++;
        }
        // fired windows form a prefix of the open ones (window ends are monotone)
        const uint64_t fire_slack = (p_.winType == TB) ? p_.triggering_delay : 0;
        while (!kd.wins.empty() && !kd.wins.front().batched && ord >= kd.wins.front().end + fire_slack) {
            fire(key, hashcode, kd, kd.wins.front());
            kd.wins.pop_front();
        }
        // windows containing the tuple are the contiguous range [first_w, last_w]
        if (!marker && !in_gap && !kd.wins.empty()) {
            uint64_t first_w;
            if (p_.win_len >= p_.slide_len)
                first_w = (rel + 1 < p_.win_len) ? 0 : (rel + 1 - p_.win_len + p_.slide_len - 1) / p_.slide_len;
            else
                first_w = (uint64_t)last_w;
            const uint64_t front_lwid = kd.wins.front().lwid;
            for (uint64_t lw = std::max(first_w, front_lwid); (long)lw <= last_w; lw++) {
                auto& w = kd.wins[lw - front_lwid];
                if (w.onOrdinal(ord, p_.winType, p_.triggering_delay) != IN) continue;
                w.no_tuples++;
                if (p_.winType == CB) w.result.setControlFields(key, w.gwid, ts_of(*t));
                if constexpr (!isNIC) meta::invoke_rich(func_, context, w.gwid, static_cast<const tuple_t&>(*t), w.result);
            }
        }
        tuple_delete(t);
    }

    // end of stream: every window still open is computed and emitted (reference :476-528)
    void flush() override {
        for (auto& kv : keys_) {
            const size_t hashcode = std::hash<key_t>()(kv.first);
            for (auto& w : kv.second.wins) fire(kv.first, hashcode, kv.second, w);
            kv.second.wins.clear();
        }
    }

    uint64_t getNumDroppedTuples() const { return dropped_; }

protected:
    void finalize_stats(Stats_Record& s) override { s.inputs_ignored = dropped_; }

private:
    struct OrdFn {
        win_type_t type;
        uint64_t operator()(const tuple_t& t) const { return type == CB ? id_of(t) : ts_of(t); }
    };
    struct KeyD {
        explicit KeyD(const WinParams& p) : archive(OrdFn{p.winType}), emit_counter(p.role == MAP ? p.map_first : 0) {}
        StreamArchive<tuple_t, OrdFn> archive;
        std::deque<Window<result_t, key_t>> wins;
        uint64_t emit_counter;
        uint64_t next_lwid = 0;
        long last_lwid = -1;
    };

    void fire(const key_t& key, size_t hashcode, KeyD& kd, Window<result_t, key_t>& w) {
        if constexpr (isNIC) {
            auto r = kd.archive.getWinRange(w.start, w.end);
            Iterable<tuple_t> iter(r.first, r.second);
            meta::invoke_rich(func_, context, w.gwid, static_cast<const Iterable<tuple_t>&>(iter), w.result);
            // tuples before the next window of this replica are no longer needed
            kd.archive.purge(w.start + p_.slide_len);
        }
        kd.last_lwid++;
        result_t* out = tuple_new<result_t>(w.result);
        if (p_.role == MAP) {
            out->setControlFields(key, kd.emit_counter, ts_of(*out));
            kd.emit_counter += p_.map_stride;
        } else if (p_.role == PLQ) {
            const auto& c = p_.config;
            const uint64_t new_id = ((c.id_inner + c.n_inner - (hashcode % c.n_inner)) % c.n_inner) + kd.emit_counter * c.n_inner;
            out->setControlFields(key, new_id, ts_of(*out));
            kd.emit_counter++;
        }
        emit(out);
    }

    F func_;
    WinParams p_;
    std::unordered_map<key_t, KeyD> keys_;
    uint64_t dropped_ = 0;
};

// Emitter + ordering used when a window engine is fed key-partitioned
// (Key_Farm, Key_FFAT, stand-alone Win_Seq): hash routing, plus the per-key last
// tuple re-sent as end-of-stream marker is not needed here because every tuple of
// a key reaches the same replica.
template <typename tuple_t>
class KF_Emitter : public rt::Node {
public:
    KF_Emitter(routing_func_t routing, size_t n_dest) : routing_(std::move(routing)), n_dest_(n_dest) {
        for (size_t i = 0; i < n_dest_; i++) add_out_port();
    }
    void svc(void* msg, int) override {
        const tuple_t* t = untag<tuple_t>(msg);
        emit_to(msg, routing_(hash_of(*t), n_dest_));
    }

private:
    routing_func_t routing_;
    size_t n_dest_;
};

// ordering rule of every window stage fed by the MultiPipe:
//   TB: TS ordering only in DETERMINISTIC mode; CB: needs DETERMINISTIC, uses
//   TS_RENUMBERING (reference pipegraph.hpp:1846-1849, :1953-1958, :2246-2248)
template <typename tuple_t>
inline std::function<rt::Node*(Mode)> window_ordering(win_type_t type, const std::string& op_name) {
    return [type, op_name](Mode m) -> rt::Node* {
        if (type == CB) {
            if (m != Mode::DETERMINISTIC)
                fatal_error(op_name + ": count-based windows cannot be used in DEFAULT (non-deterministic) mode");
            return new Ordering_Node<tuple_t>(TS_RENUMBERING);
        }
        return m == Mode::DETERMINISTIC ? new Ordering_Node<tuple_t>(TS) : nullptr;
    };
}

// Stand-alone sequential window operator (parallelism 1).  The reference exposes
// Win_Seq only as a building block with a builder (wf/builders.hpp:796); here it
// can also be added to a MultiPipe.
template <typename tuple_t, typename result_t>
class Win_Seq : public Basic_Operator {
public:
    template <typename F>
    Win_Seq(F func, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, std::string name,
            meta::closing_func_t closing = nullptr)
        : name_(std::move(name)), type_(type) {
        WinParams p;
        p.win_len = win_len;
        p.slide_len = slide_len;
        p.triggering_delay = triggering_delay;
        p.winType = type;
        p.config = WinOperatorConfig(0, 1, slide_len, 0, 1, slide_len);
        std::string n = name_;
        factory_ = [func, p, n, closing](std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new Win_Seq_Node<tuple_t, result_t, F>(func, p, n, 1, 0, closing, std::move(hub));
        };
    }
    std::string getName() const override { return name_; }
    size_t getParallelism() const override { return 1; }
    routing_modes_t getRoutingMode() const override { return COMPLEX; }
    bool isWindowed() const override { return true; }
    const std::type_info& input_type() const override { return typeid(tuple_t); }
    const std::type_info& output_type() const override { return typeid(result_t); }
    std::vector<StageSpec> build_stages(rt::Graph& g, Mode) override {
        StageSpec s;
        s.name = name_;
        s.routing = COMPLEX;
        s.replicas.push_back(g.adopt(factory_(hub_)));
        s.make_emitter = [](rt::Graph& gr, size_t) { return single_emitter(gr, new Standard_Emitter<tuple_t>(1), 1); };
        s.make_ordering = window_ordering<tuple_t>(type_, name_);
        return {s};
    }

private:
    std::string name_;
    win_type_t type_;
    std::function<rt::Node*(std::shared_ptr<StatsHub>)> factory_;
};

}  // namespace wf
