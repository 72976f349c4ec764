// Win_SeqFFAT: sequential incremental window engine for lift + combine queries,
// backed by one FlatFAT per key, and the stand-alone Win_SeqFFAT operator.
//
// Semantics: /root/reference/wf/win_seqffat.hpp:330-652, SURVEY.md A.10.
//  * count-based: lift every tuple; the first window fires when win_len lifted
//    values are present, the following ones every slide_len; on fire the pending
//    values enter the tree, the tree aggregate is the window result, the slide_len
//    oldest leaves leave the tree.
//  * time-based: time is cut in quanta of gcd(win, slide); tuples are folded into
//    per-quantum partials; a quantum is complete when its last instant plus the
//    triggering delay is older than the current tuple; complete quanta feed the
//    count-based logic with win/quantum and slide/quantum.  Tuples older than the
//    oldest open quantum are dropped.
//  * end of stream: pending values enter the tree and windows are emitted until
//    the tree is empty.
// Sliding windows only (slide < win), as in the reference (:175-178).
#pragma once
#include <numeric>
#include <unordered_map>
#include <vector>

#include "flatfat.hpp"
#include "win_seq.hpp"

namespace wf {

template <typename tuple_t, typename result_t, typename lift_t, typename comb_t>
class Win_SeqFFAT_Node : public Replica_Base {
public:
    using key_t = key_of_t<tuple_t>;

    Win_SeqFFAT_Node(lift_t lift, comb_t comb, WinParams p, std::string name, size_t par, size_t idx, meta::closing_func_t closing,
                     std::shared_ptr<StatsHub> hub)
        : Replica_Base(std::move(name), par, idx, std::move(closing), std::move(hub), true), lift_(std::move(lift)), comb_(std::move(comb)), p_(p) {
        if (p_.win_len == 0 || p_.slide_len == 0) fatal_error("window length or slide cannot be zero");
        if (p_.slide_len >= p_.win_len) fatal_error("FlatFAT-based operators can be used with sliding windows only (slide < window)");
        if (p_.winType == TB) {
            quantum_ = std::gcd(p_.win_len, p_.slide_len);
            win_q_ = p_.win_len / quantum_;
            slide_q_ = p_.slide_len / quantum_;
        } else {
            win_q_ = p_.win_len;
            slide_q_ = p_.slide_len;
        }
        set_io_bytes(sizeof(tuple_t), sizeof(result_t));
    }

    void svc(void* msg, int) override {
        tuple_t* t = untag<tuple_t>(msg);
        if (is_marker(msg)) {  // key-partitioned input: markers carry no information here
            tuple_delete(t);
            return;
        }
        const key_t key = key_of(*t);
        auto it = keys_.find(key);
        if (it == keys_.end()) it = keys_.emplace(key, KeyD(make_comb(), win_q_)).first;
        KeyD& kd = it->second;
        result_t lifted{};
        lifted.setControlFields(key, 0, ts_of(*t));
        meta::invoke_rich(lift_, context, static_cast<const tuple_t&>(*t), lifted);
        if (p_.winType == CB) {
            add_value(key, kd, lifted);
        } else {
            const uint64_t ts = ts_of(*t);
            const uint64_t q = ts / quantum_;
            if (q < kd.last_quantum) {
                dropped_++;
                tuple_delete(t);
                return;
            }
            const uint64_t distance = q - kd.last_quantum;
            while (kd.acc.size() <= distance) {
                result_t r{};
                r.setControlFields(key, kd.cb_id++, (kd.last_quantum + kd.acc.size() + 1) * quantum_ - 1);
                kd.acc.push_back(r);
                kd.acc_used.push_back(0);
            }
            if (kd.acc_used[distance]) {
                result_t tmp{};
                tmp.setControlFields(key, id_of(kd.acc[distance]), ts_of(kd.acc[distance]));
                call_comb(kd.acc[distance], lifted, tmp);
                tmp.setControlFields(key, id_of(kd.acc[distance]), ts_of(kd.acc[distance]));
                kd.acc[distance] = tmp;
            } else {
                lifted.setControlFields(key, id_of(kd.acc[distance]), ts_of(kd.acc[distance]));
                kd.acc[distance] = lifted;
                kd.acc_used[distance] = 1;
            }
            // complete quanta (respecting the triggering delay) feed the count-based logic
            size_t done = 0;
            for (size_t i = 0; i < kd.acc.size(); i++) {
                const uint64_t final_ts = (kd.last_quantum + 1) * quantum_ - 1;
                if (final_ts + p_.triggering_delay < ts) {
                    add_value(key, kd, kd.acc[i]);
                    kd.last_quantum++;
                    done++;
                } else {
                    break;
                }
            }
            kd.acc.erase(kd.acc.begin(), kd.acc.begin() + (std::ptrdiff_t)done);
            kd.acc_used.erase(kd.acc_used.begin(), kd.acc_used.begin() + (std::ptrdiff_t)done);
        }
        tuple_delete(t);
    }

    void flush() override {
        for (auto& kv : keys_) {
            KeyD& kd = kv.second;
            for (size_t i = 0; i < kd.acc.size(); i++) {
                add_value(kv.first, kd, kd.acc[i]);
                kd.last_quantum++;
            }
            kd.acc.clear();
            kd.acc_used.clear();
            kd.fat.insert(kd.pending);
            kd.pending.clear();
            while (!kd.fat.is_Empty()) emit_window(kv.first, kd);
        }
    }

    uint64_t getNumDroppedTuples() const { return dropped_; }

protected:
    void finalize_stats(Stats_Record& s) override { s.inputs_ignored = dropped_; }

private:
    using fat_comb_t = std::function<void(const result_t&, const result_t&, result_t&)>;
    struct KeyD {
        KeyD(fat_comb_t c, size_t win) : fat(std::move(c), win) {}
        FlatFAT<result_t, fat_comb_t> fat;
        std::vector<result_t> pending;
        std::vector<result_t> acc;  // TB: partials of the open quanta
        std::vector<uint8_t> acc_used;
        uint64_t cb_id = 0;
        uint64_t last_quantum = 0;
        uint64_t rcv_counter = 0;
        uint64_t slide_counter = 0;
        uint64_t next_lwid = 0;
    };

    fat_comb_t make_comb() {
        return [this](const result_t& a, const result_t& b, result_t& out) { call_comb(a, b, out); };
    }
    void call_comb(const result_t& a, const result_t& b, result_t& out) { meta::invoke_rich(comb_, context, a, b, out); }

    // count-based firing logic over lifted values / complete quanta
    void add_value(const key_t& key, KeyD& kd, const result_t& v) {
        kd.pending.push_back(v);
        kd.rcv_counter++;
        kd.slide_counter++;
        bool fired = false;
        if (kd.rcv_counter == win_q_) {
            fired = true;
        } else if (kd.rcv_counter > win_q_ && kd.slide_counter % slide_q_ == 0) {
            fired = true;
        }
        if (fired) {
            kd.slide_counter = 0;
            kd.fat.insert(kd.pending);
            kd.pending.clear();
            emit_window(key, kd);
        }
    }
    void emit_window(const key_t& key, KeyD& kd) {
        const WinAssign wa(p_.config, std::hash<key_t>()(key), p_.role);
        const uint64_t gwid = wa.first_gwid_key + kd.next_lwid * wa.gwid_stride;
        kd.next_lwid++;
        result_t* out = tuple_new<result_t>();
        kd.fat.getResult(*out);
        kd.fat.remove(slide_q_);
        out->setControlFields(key, gwid, ts_of(*out));
        emit(out);
    }

    lift_t lift_;
    comb_t comb_;
    WinParams p_;
    uint64_t quantum_ = 1, win_q_ = 1, slide_q_ = 1;
    std::unordered_map<key_t, KeyD> keys_;
    uint64_t dropped_ = 0;
};

// Stand-alone operator, parallelism 1 (builder: WinSeqFFAT_Builder).
template <typename tuple_t, typename result_t>
class Win_SeqFFAT : public Basic_Operator {
public:
    template <typename lift_t, typename comb_t>
    Win_SeqFFAT(lift_t lift, comb_t comb, uint64_t win_len, uint64_t slide_len, uint64_t triggering_delay, win_type_t type, std::string name,
                meta::closing_func_t closing = nullptr)
        : name_(std::move(name)), type_(type) {
        WinParams p;
        p.win_len = win_len;
        p.slide_len = slide_len;
        p.triggering_delay = triggering_delay;
        p.winType = type;
        p.config = WinOperatorConfig(0, 1, slide_len, 0, 1, slide_len);
        std::string n = name_;
        factory_ = [lift, comb, p, n, closing](std::shared_ptr<StatsHub> hub) -> rt::Node* {
            return new Win_SeqFFAT_Node<tuple_t, result_t, lift_t, comb_t>(lift, comb, p, n, 1, 0, closing, std::move(hub));
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
