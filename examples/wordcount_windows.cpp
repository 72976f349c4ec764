// CPU example: sliding-window word count.
//   Source (sentences) -> FlatMap (split into words) -> Key_Farm (count per word, TB window 1 s / slide 250 ms) -> Sink
// build:  g++ -O2 -std=c++17 -I windflow_amd/include examples/wordcount_windows.cpp -o wordcount -pthread
#include <iostream>
#include <optional>
#include <sstream>
#include <string>
#include <tuple>

#include <wf/windflow.hpp>

using namespace wf;

struct Sentence {
    std::string text;
    uint64_t id = 0, ts = 0;
    std::tuple<std::string, uint64_t, uint64_t> getControlFields() const { return {"", id, ts}; }
    void setControlFields(std::string, uint64_t i, uint64_t t) {
        id = i;
        ts = t;
    }
};
struct Word {
    std::string word;
    uint64_t id = 0, ts = 0;
    uint64_t count = 0;
    std::tuple<std::string, uint64_t, uint64_t> getControlFields() const { return {word, id, ts}; }
    void setControlFields(std::string w, uint64_t i, uint64_t t) {
        word = std::move(w);
        id = i;
        ts = t;
    }
};

int main() {
    const char* corpus[] = {"the quick brown fox", "jumps over the lazy dog", "the dog barks", "a quick reply"};
    PipeGraph graph("wordcount", Mode::DETERMINISTIC);
    uint64_t n = 0;
    Source src = Source_Builder([&](Sentence& s) {
                     s.text = corpus[n % 4];
                     s.id = n;
                     s.ts = n * 50000;  // one sentence every 50 ms of event time
                     return ++n < 200;
                 }).withName("sentences").build();
    MultiPipe& mp = graph.add_source(src);
    FlatMap split = FlatMap_Builder([](const Sentence& s, Shipper<Word>& out) {
                        std::istringstream in(s.text);
                        std::string w;
                        while (in >> w) {
                            Word x;
                            x.setControlFields(w, s.id, s.ts);
                            out.push(std::move(x));
                        }
                    }).withName("splitter").withParallelism(2).build();
    mp.add(split);
    Key_Farm counter = KeyFarm_Builder([](uint64_t, const Word&, Word& result) { result.count++; })
                           .withTBWindows(std::chrono::seconds(1), std::chrono::milliseconds(250))
                           .withName("counter").withParallelism(3).build();
    mp.add(counter);
    Sink sink = Sink_Builder([](std::optional<Word>& w) {
                    if (w && w->word == "the") std::cout << "window " << w->id << " [..." << w->ts << " us]: the x" << w->count << std::endl;
                }).withName("printer").build();
    mp.add_sink(sink);
    graph.run();
    return 0;
}
