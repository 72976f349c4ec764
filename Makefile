# Native build without Python: the library is header-only, this builds the regression programs,
# the applications and the examples.   make [all|cpu|gpu|clean]   (GPU targets need hipcc, gfx950)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
INC      := -Iwindflow_amd/include -Iwindflow_amd/csrc -Itests/cpp
CXXFLAGS := -O2 -std=c++17 -Wall -Wextra $(INC)
# -unroll-runtime: see windflow_amd/build.py DEVICE_FLAGS
HIPFLAGS := --offload-arch=$(ARCH) -O2 -std=c++17 -mllvm -unroll-runtime $(INC)

CPU_BINS := tests/cpp/bin/smoke_basic_ops tests/cpp/bin/test_mp_windows tests/cpp/bin/test_graphs tests/cpp/bin/test_runtime \
            tests/cpp/bin/test_api_variants apps/bin/ysb_kf apps/bin/ysb_wmr apps/bin/bench_cpu apps/bin/example_wordcount
GPU_BINS := tests/cpp/bin/test_mp_windows_gpu tests/cpp/bin/test_mp_windows_gpu_fixedkey tests/cpp/bin/test_gpu_keyed apps/bin/ysb_gpu apps/bin/bench_ops_gpu apps/bin/bench_win_gpu \
            apps/bin/example_gpu_pipeline apps/bin/example_gpu_pane_farm
HEADERS  := $(shell find windflow_amd/include windflow_amd/csrc tests/cpp apps/ysb -name '*.hpp')

all: cpu gpu
cpu: $(CPU_BINS)
gpu: $(GPU_BINS)

tests/cpp/bin/%: tests/cpp/%.cpp $(HEADERS)
	@mkdir -p $(@D)
	$(CXX) $(CXXFLAGS) $< -o $@ -pthread
tests/cpp/bin/%: tests/cpp/%.hip $(HEADERS)
	@mkdir -p $(@D)
	$(HIPCC) $(HIPFLAGS) $< -o $@ -lpthread
# the window suite keyed by 16-byte fixed-size keys (the reference's *_string programs)
tests/cpp/bin/test_mp_windows_gpu_fixedkey: tests/cpp/test_mp_windows_gpu.hip $(HEADERS)
	@mkdir -p $(@D)
	$(HIPCC) $(HIPFLAGS) -DWF_TEST_FIXED_KEY $< -o $@ -lpthread
apps/bin/%: apps/ysb/%.cpp $(HEADERS)
	@mkdir -p $(@D)
	$(CXX) $(CXXFLAGS) -Iapps/ysb $< -o $@ -pthread
apps/bin/%: apps/ysb/%.hip $(HEADERS)
	@mkdir -p $(@D)
	$(HIPCC) $(HIPFLAGS) -Iapps/ysb $< -o $@ -lpthread
apps/bin/%: apps/bench/%.cpp $(HEADERS)
	@mkdir -p $(@D)
	$(CXX) $(CXXFLAGS) $< -o $@ -pthread
apps/bin/%: apps/bench/%.hip $(HEADERS)
	@mkdir -p $(@D)
	$(HIPCC) $(HIPFLAGS) $< -o $@ -lpthread
apps/bin/example_wordcount: examples/wordcount_windows.cpp $(HEADERS)
	@mkdir -p $(@D)
	$(CXX) $(CXXFLAGS) $< -o $@ -pthread
apps/bin/example_gpu_pipeline: examples/gpu_pipeline.hip $(HEADERS)
	@mkdir -p $(@D)
	$(HIPCC) $(HIPFLAGS) $< -o $@ -lpthread
apps/bin/example_gpu_pane_farm: examples/gpu_pane_farm.hip $(HEADERS)
	@mkdir -p $(@D)
	$(HIPCC) $(HIPFLAGS) $< -o $@ -lpthread

# ThreadSanitizer build of the runtime unit tests (host code only)
tsan: tests/cpp/test_runtime.cpp $(HEADERS)
	@mkdir -p tests/cpp/bin
	$(CXX) -O1 -g -std=c++17 -fsanitize=thread $(INC) $< -o tests/cpp/bin/test_runtime_tsan -pthread
	WF_QUIET=1 tests/cpp/bin/test_runtime_tsan

# AddressSanitizer + UndefinedBehaviorSanitizer builds of the CPU regression programs (host code only)
ASAN_PROGS := test_runtime test_api_variants test_mp_windows test_graphs
asan: $(HEADERS)
	@mkdir -p tests/cpp/bin
	for p in $(ASAN_PROGS); do $(CXX) -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
	    $(INC) tests/cpp/$$p.cpp -o tests/cpp/bin/$${p}_asan -pthread || exit 1; done
	WF_QUIET=1 tests/cpp/bin/test_runtime_asan && WF_QUIET=1 tests/cpp/bin/test_api_variants_asan && \
	WF_QUIET=1 tests/cpp/bin/test_mp_windows_asan -r 1 -l 300 -k 3 -w 9 -s 3 && WF_QUIET=1 tests/cpp/bin/test_graphs_asan -r 1

check-cpu: cpu
	WF_QUIET=1 tests/cpp/bin/test_runtime && WF_QUIET=1 tests/cpp/bin/test_api_variants && WF_QUIET=1 tests/cpp/bin/test_graphs -r 1

clean:
	rm -rf tests/cpp/bin apps/bin

.PHONY: all cpu gpu tsan asan check-cpu clean
