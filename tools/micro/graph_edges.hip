// Fork/join edges of a three-stream MD step: events between streams against a captured HIP graph.
// The step's shape is the engine's plain step with stand-in kernels that spin for a fixed time (100 MHz wall clock):
//   main stream:   nonbonded (100 us, 256 workgroups of 768)            -> join -> update (16 us)
//   PME stream:    spread (30) -> fft (12) -> fft (25) -> fft (8) -> gather (11)     (forked behind the update)
//   bonded stream: bonded (6)
// Ideal step: 116 us (nonbonded + update; the PME chain is shorter). What is measured is the step period over many steps
// with the dependencies expressed (a) as the engine does it: the update's and the gather's completion signals as events
// (hipExtLaunchKernelGGL), hipStreamWaitEvent for fork and join; (b) with marker events recorded behind the kernels
// (hipEventRecord); (c) as a graph captured from (b) over `NS` steps and launched repeatedly.
// Question: do a graph's edges cost less than the 4-8 us a stream pays for a wait on another stream's event?
// Build: hipcc -O3 --offload-arch=gfx950 -o graph_edges graph_edges.hip ; run: ./graph_edges
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_spin(long long ticks, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

struct Streams { hipStream_t main, pme, bond; hipEvent_t fork, pme_done, bond_done; int *sink; };

static void spin(hipStream_t st, int blocks, int threads, double us, int *sink, hipEvent_t stop = nullptr)
{
    const long long ticks = (long long)(us * 100.0);
    if (stop) hipExtLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), 0, st, nullptr, stop, 0, ticks, sink);
    else hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), 0, st, ticks, sink);
}

// one step; `signals`: completion signals of the kernels themselves serve as events (not capturable)
static void step(Streams &S, bool signals)
{
    // side streams were released by the previous update (or by the caller for the first step)
    spin(S.main, 256, 768, 100.0, S.sink);
    spin(S.pme, 560, 256, 30.0, S.sink); spin(S.pme, 280, 256, 12.0, S.sink); spin(S.pme, 41, 512, 25.0, S.sink); spin(S.pme, 280, 256, 8.0, S.sink);
    if (signals) spin(S.pme, 1024, 256, 11.0, S.sink, S.pme_done);
    else { spin(S.pme, 1024, 256, 11.0, S.sink); CHECK(hipEventRecord(S.pme_done, S.pme)); }
    spin(S.bond, 64, 256, 6.0, S.sink);
    CHECK(hipEventRecord(S.bond_done, S.bond));
    CHECK(hipStreamWaitEvent(S.main, S.bond_done, 0));
    CHECK(hipStreamWaitEvent(S.main, S.pme_done, 0));
    if (signals) spin(S.main, 1100, 128, 16.0, S.sink, S.fork);
    else { spin(S.main, 1100, 128, 16.0, S.sink); CHECK(hipEventRecord(S.fork, S.main)); }
    CHECK(hipStreamWaitEvent(S.pme, S.fork, 0));
    CHECK(hipStreamWaitEvent(S.bond, S.fork, 0));
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    Streams S;
    CHECK(hipStreamCreate(&S.main)); CHECK(hipStreamCreate(&S.pme)); CHECK(hipStreamCreate(&S.bond));
    CHECK(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&S.pme_done, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&S.bond_done, hipEventDisableTiming));
    CHECK(hipMalloc(&S.sink, sizeof(int))); CHECK(hipMemset(S.sink, 0, sizeof(int)));
    const int NS = 8, REP = 50;
    for (int mode = 0; mode < 2; mode++) {
        for (int pass = 0; pass < 2; pass++) {       // pass 0 warms up
            CHECK(hipDeviceSynchronize());
            const double t0 = now_us();
            for (int i = 0; i < NS * REP; i++) step(S, mode == 0);
            CHECK(hipDeviceSynchronize());
            if (pass) printf("%-46s %7.1f us per step (ideal 116)\n", mode == 0 ? "events: kernel completion signals" : "events: markers behind the kernels", (now_us() - t0) / (NS * REP));
        }
    }
    // graph: NS steps captured from the marker form; the side streams join the origin stream before the capture ends
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(S.main, hipStreamCaptureModeGlobal));
    CHECK(hipEventRecord(S.fork, S.main));
    CHECK(hipStreamWaitEvent(S.pme, S.fork, 0)); CHECK(hipStreamWaitEvent(S.bond, S.fork, 0));
    for (int i = 0; i < NS; i++) step(S, false);
    CHECK(hipEventRecord(S.pme_done, S.pme)); CHECK(hipEventRecord(S.bond_done, S.bond));
    CHECK(hipStreamWaitEvent(S.main, S.pme_done, 0)); CHECK(hipStreamWaitEvent(S.main, S.bond_done, 0));
    CHECK(hipStreamEndCapture(S.main, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    size_t nn = 0; CHECK(hipGraphGetNodes(graph, nullptr, &nn));
    for (int pass = 0; pass < 2; pass++) {
        CHECK(hipDeviceSynchronize());
        const double t0 = now_us();
        for (int i = 0; i < REP; i++) CHECK(hipGraphLaunch(exec, S.main));
        CHECK(hipDeviceSynchronize());
        if (pass) printf("%-46s %7.1f us per step (ideal 116; %zu nodes per graph of %d steps)\n", "graph captured from the marker form", (now_us() - t0) / (NS * REP), nn, NS);
    }
    int h = 0; CHECK(hipMemcpy(&h, S.sink, sizeof(int), hipMemcpyDeviceToHost));
    printf("kernels run: %d\n", h);
    return 0;
}
