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
// Per scheme the gaps of the critical path are printed from time stamps the stand-in kernels take themselves.
// Build: hipcc -O3 --offload-arch=gfx950 -o graph_edges graph_edges.hip ; run: ./graph_edges
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// stamps (100 MHz ticks): slot 2k first start (block 0), slot 2k + 1 last end (every block: max)
__global__ void k_spin(long long ticks, int *sink, long long *stamp)
{
    const long long t0 = wall_clock64();
    if (stamp && threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (stamp && threadIdx.x == 0 && (blockIdx.x & 63) == 63) atomicMax((unsigned long long *)&stamp[1], (unsigned long long)wall_clock64());     // (a sample of the blocks: atomics on one address retire one after the other)
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

struct Streams { hipStream_t main, pme, bond; hipEvent_t fork, pme_done, bond_done; int *sink; long long *stamps; int nstep; };
enum { K_NB = 0, K_GATHER = 1, K_UPDATE = 2, K_SPREAD = 3, NK = 4 };

static void spin(hipStream_t st, int blocks, int threads, double us, int *sink, hipEvent_t stop = nullptr, long long *stamp = nullptr)
{
    const long long ticks = (long long)(us * 100.0);
    if (stop) hipExtLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), 0, st, nullptr, stop, 0, ticks, sink, stamp);
    else hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), 0, st, ticks, sink, stamp);
}

// one step. mode 0: completion signals of the update and the gather serve as events (the engine's scheme; not capturable);
// 1: marker events recorded behind the kernels; 2: as 0, but the update is a plain launch and the fork is a marker behind it
// (is it the completion signal on the update that holds the next nonbonded launch back?); 3: as 0, but the join waits for
// one event only (the bonded kernel joins the PME stream ahead of the gather)
static double g_side = 1.0;       // scale of the side streams' kernel times (argv[1]): with 0.5 the PME chain ends long before the nonbonded kernel
static void step(Streams &S, int mode)
{
    long long *st = S.stamps ? S.stamps + (size_t)S.nstep * NK * 2 : nullptr;
    S.nstep++;
    const bool sig = mode == 0 || mode == 2 || mode == 3;
    // side streams were released by the previous update (or by the caller for the first step)
    spin(S.main, 256, 768, 100.0, S.sink, nullptr, st ? st + 2 * K_NB : nullptr);
    spin(S.pme, 560, 256, 30.0 * g_side, S.sink, nullptr, st ? st + 2 * K_SPREAD : nullptr); spin(S.pme, 280, 256, 12.0 * g_side, S.sink); spin(S.pme, 41, 512, 25.0 * g_side, S.sink); spin(S.pme, 280, 256, 8.0 * g_side, S.sink);
    spin(S.bond, 64, 256, 6.0, S.sink);
    CHECK(hipEventRecord(S.bond_done, S.bond));
    if (mode == 3) CHECK(hipStreamWaitEvent(S.pme, S.bond_done, 0));
    if (sig) spin(S.pme, 1024, 256, 11.0 * g_side, S.sink, S.pme_done, st ? st + 2 * K_GATHER : nullptr);
    else { spin(S.pme, 1024, 256, 11.0 * g_side, S.sink, nullptr, st ? st + 2 * K_GATHER : nullptr); CHECK(hipEventRecord(S.pme_done, S.pme)); }
    if (mode != 3) CHECK(hipStreamWaitEvent(S.main, S.bond_done, 0));
    CHECK(hipStreamWaitEvent(S.main, S.pme_done, 0));
    if (mode == 0 || mode == 3) spin(S.main, 1100, 128, 16.0, S.sink, S.fork, st ? st + 2 * K_UPDATE : nullptr);
    else { spin(S.main, 1100, 128, 16.0, S.sink, nullptr, st ? st + 2 * K_UPDATE : nullptr); CHECK(hipEventRecord(S.fork, S.main)); }
    CHECK(hipStreamWaitEvent(S.pme, S.fork, 0));
    CHECK(hipStreamWaitEvent(S.bond, S.fork, 0));
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc > 1) g_side = atof(argv[1]);
    printf("side-stream kernel times x %.2f\n", g_side);
    Streams S;
    CHECK(hipStreamCreate(&S.main)); CHECK(hipStreamCreate(&S.pme)); CHECK(hipStreamCreate(&S.bond));
    CHECK(hipEventCreateWithFlags(&S.fork, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&S.pme_done, hipEventDisableTiming));
    CHECK(hipEventCreateWithFlags(&S.bond_done, hipEventDisableTiming));
    CHECK(hipMalloc(&S.sink, sizeof(int))); CHECK(hipMemset(S.sink, 0, sizeof(int)));
    const int NS = 8, REP = 50;
    const size_t nst = (size_t)NS * REP * NK * 2;
    CHECK(hipMalloc(&S.stamps, sizeof(long long) * nst));
    long long *h = (long long *)malloc(sizeof(long long) * nst);
    const char *names[4] = { "events: kernel completion signals", "events: markers behind the kernels", "signals, but the fork a marker behind the update",
                             "signals, bonded joins through the PME stream" };
    for (int mode = 0; mode < 4; mode++) {
        for (int pass = 0; pass < 2; pass++) {       // pass 0 warms up
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemset(S.stamps, 0, sizeof(long long) * nst)); S.nstep = 0;
            const double t0 = now_us();
            for (int i = 0; i < NS * REP; i++) step(S, mode);
            CHECK(hipDeviceSynchronize());
            if (!pass) continue;
            const double per = (now_us() - t0) / (NS * REP);
            CHECK(hipMemcpy(h, S.stamps, sizeof(long long) * nst, hipMemcpyDeviceToHost));
            double g_un = 0, g_nu = 0, g_gu = 0, g_us = 0; int n = 0;
            for (int i = 10; i + 1 < NS * REP; i++, n++) {
                const long long *a = h + (size_t)i * NK * 2, *b = a + NK * 2;
                g_nu += (a[2 * K_UPDATE] - a[2 * K_NB + 1]) / 100.0;       // nonbonded end -> update start
                g_gu += (a[2 * K_UPDATE] - a[2 * K_GATHER + 1]) / 100.0;   // gather end -> update start
                g_un += (b[2 * K_NB] - a[2 * K_UPDATE + 1]) / 100.0;       // update end -> next nonbonded start
                g_us += (b[2 * K_SPREAD] - a[2 * K_UPDATE + 1]) / 100.0;   // update end -> next spread start (fork)
            }
            printf("%-50s %6.1f us per step | nonbonded end -> update %5.1f, gather end -> update %5.1f, update end -> nonbonded %5.1f, update end -> spread %5.1f\n",
                   names[mode], per, g_nu / n, g_gu / n, g_un / n, g_us / n);
        }
    }
    S.stamps = nullptr;
    // graph: NS steps captured from the marker form; the side streams join the origin stream before the capture ends
    hipGraph_t graph; hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(S.main, hipStreamCaptureModeGlobal));
    CHECK(hipEventRecord(S.fork, S.main));
    CHECK(hipStreamWaitEvent(S.pme, S.fork, 0)); CHECK(hipStreamWaitEvent(S.bond, S.fork, 0));
    for (int i = 0; i < NS; i++) step(S, 1);
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
    int hs = 0; CHECK(hipMemcpy(&hs, S.sink, sizeof(int), hipMemcpyDeviceToHost));
    printf("kernels run: %d\n", hs);
    return 0;
}
