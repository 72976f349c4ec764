#include <wf/windflow.hpp>
#include <atomic>
using namespace wf; using namespace std;
struct tuple_t { size_t key; uint64_t id, ts; int64_t value;
  tuple<size_t,uint64_t,uint64_t> getControlFields() const { return {key,id,ts}; }
  void setControlFields(size_t k, uint64_t i, uint64_t t){key=k;id=i;ts=t;} };
atomic<long> gsum{0};
int main(){
  for (int run=0; run<5; run++){
  gsum=0;
  PipeGraph graph("t1", run%2? Mode::DETERMINISTIC: Mode::DEFAULT);
  size_t n=0; 
  auto src_f=[n](tuple_t& t) mutable { t.key=n%7; t.id=n/7; t.ts=n; t.value=n; n++; return n<100000; };
  Source source = Source_Builder(src_f).withName("src").withParallelism(1+run%2).build();
  MultiPipe& mp = graph.add_source(source);
  Filter filter = Filter_Builder([](tuple_t& t){return t.value%2==0;}).withParallelism(1+run).build();
  mp.chain(filter);
  FlatMap fm = FlatMap_Builder([](const tuple_t& t, Shipper<tuple_t>& s){ for(int i=0;i<3;i++){tuple_t x=t; x.value+=i; s.push(x);} }).withParallelism(2).build();
  mp.add(fm);
  Map map = Map_Builder([](tuple_t& t){t.value*=2;}).withParallelism(2).build();
  mp.chain(map);
  Accumulator acc = Accumulator_Builder([](const tuple_t& t, tuple_t& r){ r.key=t.key; r.value+=t.value; }).withParallelism(3).build();
  mp.add(acc);
  Sink sink = Sink_Builder([](optional<tuple_t>& o){ if(o) gsum+= 1; }).withParallelism(2).build();
  mp.add_sink(sink);
  printf("threads=%zu\n", graph.getNumThreads());
  graph.run();
  printf("sum=%ld\n", gsum.load());
  }
}
