// Run control around the step loop: start-up (velocity generation, constraining the start
// configuration), continuation (step counter), steepest-descent minimisation and asynchronous
// trajectory-frame capture.
// Replaces what `gen_vel/gen_temp/gen_seed` (/root/reference/mdp/nvt_complex.mdp:39-41),
// `continuation = no` (nvt_complex.mdp:13), `integrator = steep`, `emtol`, `emstep`
// (/root/reference/mdp/em_complex.mdp:3-6) and the output cadence keys `nstxout`,
// `nstvout`, `nstxout-compressed` (/root/reference/mdp/production.mdp:7-12) switch on in gmx mdrun.
#include "mdx_internal.h"
#include <chrono>
#include <climits>
#include <mutex>
#include <thread>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cmath>
#include <cstring>

#define ARG_CHECK(sys, cond, msg)                                   \
    do {                                                            \
        if (!(cond)) { (sys)->err = (msg); return MDX_ERR_ARG; }    \
    } while (0)

extern "C" int mdx_set_step(mdx_system *s, long long step)
{
    ARG_CHECK(s, step >= 0, "mdx_set_step: step must be >= 0");
    s->step = step;
    s->steps_since_search = 1 << 30;
    s->first_step = true;
    return MDX_OK;
}

// ---------------------------------------------------------------- velocity generation
// Maxwell-Boltzmann: each component N(0, kT/m), one counter-based stream per atom.
__global__ __launch_bounds__(256) void k_gen_vel(int n, float4 *__restrict__ v, double kT, unsigned long long seed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 vi = v[i];
    if (vi.w > 0.0f) {
        RngStream rng(seed, 0x9e0000ull, (uint64_t)i);
        const double sd = sqrt(kT * (double)vi.w);
        vi.x = (float)(sd * rng.gauss()); vi.y = (float)(sd * rng.gauss()); vi.z = (float)(sd * rng.gauss());
    } else {
        vi.x = vi.y = vi.z = 0.0f;
    }
    v[i] = vi;
}

__global__ __launch_bounds__(256) void k_scale_vel_to(int n, float4 *__restrict__ v, const DevState *__restrict__ d, double ekin_target)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double ek = 0.0;
    for (int g = 0; g < MDX_MAX_TC; g++) ek += d->ekinh[g];
    const float sc = ek > 0.0 ? (float)sqrt(ekin_target / ek) : 1.0f;
    float4 vi = v[i];
    vi.x *= sc; vi.y *= sc; vi.z *= sc;
    v[i] = vi;
}

extern "C" int mdx_generate_velocities(mdx_system *s, double temperature, unsigned long long seed)
{
    if (s->frame_ring) { const int rc = mdx_frames_flush(s, s->stream); if (rc) return rc; }
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, temperature >= 0.0, "mdx_generate_velocities: negative temperature");
    const int n = s->n, nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_gen_vel, dim3(nb), dim3(256), 0, s->stream, n, s->v, MDX_BOLTZ * temperature, seed);
    int rc;
    if ((rc = mdx_launch_comm(s))) return rc;
    // scale to exactly `temperature` counting 3N degrees of freedom, as the generator does before
    // any constraint is applied
    if ((rc = mdx_launch_clear(s, false, true, false))) return rc;
    if ((rc = mdx_launch_ke(s, false, false))) return rc;
    hipLaunchKernelGGL(k_scale_vel_to, dim3(nb), dim3(256), 0, s->stream, n, s->v, s->d, 0.5 * 3.0 * n * MDX_BOLTZ * temperature);
    MDX_CHECK(s, hipGetLastError());
    s->first_step = true;
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    return MDX_OK;
}

// ---------------------------------------------------------------- constraining the start
__global__ __launch_bounds__(256) void k_copy_pos(int n, const float4 *__restrict__ x, float4 *__restrict__ xp, const float4 *__restrict__ v, float back_dt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = x[i];
    if (back_dt != 0.0f) { const float4 w = v[i]; p.x -= back_dt * w.x; p.y -= back_dt * w.y; p.z -= back_dt * w.z; }
    xp[i] = p;
}

extern "C" int mdx_constrain_start(mdx_system *s)
{
    if (s->frame_ring) { const int rc = mdx_frames_flush(s, s->stream); if (rc) return rc; }
    (void)hipSetDevice(s->device);
    if (s->nsettle == 0 && s->ncomp == 0) return MDX_OK;
    const int n = s->n, nb = (n + 255) / 256;
    int rc;
    // positions: project x(0) onto the constraint manifold, reference directions from x(0) itself
    hipLaunchKernelGGL(k_copy_pos, dim3(nb), dim3(256), 0, s->stream, n, s->x, s->xp, s->v, 0.0f);
    if ((rc = mdx_launch_constrain(s, 0.0f, false, std::max(s->lincs_iter, 4)))) return rc;
    { float4 *t = s->x; s->x = s->xp; s->xp = t; }
    // velocities: constrain x(-dt) = x(0) - dt v against x(0); the displacement it takes, divided
    // by -dt, removes the velocity components along the constraints
    hipLaunchKernelGGL(k_copy_pos, dim3(nb), dim3(256), 0, s->stream, n, s->x, s->xp, s->v, s->dt);
    if ((rc = mdx_launch_constrain(s, -1.0f / s->dt, false, std::max(s->lincs_iter, 4)))) return rc;
    MDX_CHECK(s, hipGetLastError());
    s->steps_since_search = 1 << 30; s->first_step = true; s->forces_current = false;
    MDX_CHECK(s, hipStreamSynchronize(s->stream));
    return MDX_OK;
}

// ---------------------------------------------------------------- steepest descent
__device__ inline void atomic_max_pos_float(float *addr, float val)
{
    atomicMax((int *)addr, __float_as_int(val));      // valid for non-negative floats
}

__global__ __launch_bounds__(256) void k_fmax(int n, const float *__restrict__ f, const float *__restrict__ f_pme, const float *__restrict__ f_s,
                                              const int *__restrict__ sorted_of, const float4 *__restrict__ v, float *__restrict__ out2)
{
    __shared__ float red[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float f2 = 0.0f;
    if (i < n && v[i].w > 0.0f) {
        const int g = sorted_of[i];
        const float a = f[3 * i] + f_pme[3 * i] + f_s[3 * g], b = f[3 * i + 1] + f_pme[3 * i + 1] + f_s[3 * g + 1],
                    c = f[3 * i + 2] + f_pme[3 * i + 2] + f_s[3 * g + 2];
        f2 = a * a + b * b + c * c;
    }
    for (int o = 32; o > 0; o >>= 1) f2 = fmaxf(f2, __shfl_xor(f2, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = f2;
    __syncthreads();
    if (threadIdx.x == 0) atomic_max_pos_float(out2, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

__global__ __launch_bounds__(256) void k_sd_move(int n, const float4 *__restrict__ x, float4 *__restrict__ xp, const float *__restrict__ f,
                                                 const float *__restrict__ f_pme, const float *__restrict__ f_s,
                                                 const int *__restrict__ sorted_of, const float4 *__restrict__ v, float scale)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = x[i];
    if (v[i].w > 0.0f) {
        const int g = sorted_of[i];
        p.x += scale * (f[3 * i] + f_pme[3 * i] + f_s[3 * g]);
        p.y += scale * (f[3 * i + 1] + f_pme[3 * i + 1] + f_s[3 * g + 1]);
        p.z += scale * (f[3 * i + 2] + f_pme[3 * i + 2] + f_s[3 * g + 2]);
    }
    xp[i] = p;
}

// result6: [0] steps done, [1] converged (1) / step limit (0) / step size underflow (2),
// [2] final potential energy, [3] final max force, [4] first potential energy, [5] first max force.
// Each trial costs one full force evaluation including a new pair list (`nstlist = 1` in em.mdp).
extern "C" int mdx_minimize(mdx_system *s, long long nsteps, double emtol, double emstep, double *result6,
                            void (*progress)(long long step, double epot, double fmax, double stepsize))
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, nsteps >= 0 && emstep > 0.0, "mdx_minimize: nsteps >= 0 and emstep > 0 required");
    if (s->frame_ring) { const int rcf = mdx_frames_flush(s, s->stream); if (rcf) return rcf; }
    const int n = s->n, nb = (n + 255) / 256;
    const unsigned terms = MDX_F_ALL & ~MDX_F_PULL;
    float *d_fmax = nullptr;
    MDX_CHECK(s, hipMalloc((void **)&d_fmax, sizeof(float)));
    auto evaluate = [&](double *epot, double *fmax) -> int {
        int rc = mdx_compute_forces(s, terms);
        if (rc) return rc;
        MDX_CHECK(s, hipMemsetAsync(d_fmax, 0, sizeof(float), s->stream));
        hipLaunchKernelGGL(k_fmax, dim3(nb), dim3(256), 0, s->stream, n, s->f, s->f_pme, s->f_s, s->sorted_of, s->v, d_fmax);
        float f2 = 0.0f;
        MDX_CHECK(s, hipMemcpyAsync(&f2, d_fmax, sizeof(float), hipMemcpyDeviceToHost, s->stream));
        MDX_CHECK(s, hipStreamSynchronize(s->stream));
        *epot = s->h->ener[MDX_E_POT];
        *fmax = std::sqrt((double)f2);
        return MDX_OK;
    };
    // start from a constrained geometry
    int rc = MDX_OK;
    if (s->nsettle > 0 || s->ncomp > 0) {
        hipLaunchKernelGGL(k_copy_pos, dim3(nb), dim3(256), 0, s->stream, n, s->x, s->xp, s->v, 0.0f);
        rc = mdx_launch_constrain(s, 0.0f, false, std::max(s->lincs_iter, 4));
        if (rc == MDX_OK) { float4 *t = s->x; s->x = s->xp; s->xp = t; }
    }
    double epot = 0, fmax = 0, step = emstep;
    if (rc == MDX_OK) rc = evaluate(&epot, &fmax);
    const double e0 = epot, f0 = fmax;
    long long it = 0;
    int status = 0;
    while (rc == MDX_OK) {
        if (progress) progress(it, epot, fmax, step);
        if (!std::isfinite(epot) || !std::isfinite(fmax)) { s->err = "mdx_minimize: non-finite energy or force"; rc = MDX_ERR_STATE; break; }
        if (fmax < emtol) { status = 1; break; }
        if (it >= nsteps) { status = 0; break; }
        if (step < 1e-7 * emstep || fmax == 0.0) { status = 2; break; }
        // trial move along the force, largest displacement = step
        hipLaunchKernelGGL(k_sd_move, dim3(nb), dim3(256), 0, s->stream, n, s->x, s->xp, s->f, s->f_pme, s->f_s, s->sorted_of, s->v,
                           (float)(step / fmax));
        if ((rc = mdx_launch_constrain(s, 0.0f, false, std::max(s->lincs_iter, 2)))) break;
        { float4 *t = s->x; s->x = s->xp; s->xp = t; }      // trial becomes current, old kept in xp
        double e_try = 0, f_try = 0;
        if ((rc = evaluate(&e_try, &f_try))) break;
        it++;
        if (std::isfinite(e_try) && e_try < epot) {
            epot = e_try; fmax = f_try; step *= 1.2;
        } else {
            { float4 *t = s->x; s->x = s->xp; s->xp = t; }  // reject: restore
            step *= 0.5;
            double e_back = 0, f_back = 0;
            if ((rc = evaluate(&e_back, &f_back))) break;   // forces of the kept geometry are needed again
            epot = e_back; fmax = f_back;
        }
    }
    (void)hipFree(d_fmax);
    if (result6) { result6[0] = (double)it; result6[1] = status; result6[2] = epot; result6[3] = fmax; result6[4] = e0; result6[5] = f0; }
    s->steps_since_search = 1 << 30; s->first_step = true;
    return rc;
}

// ---------------------------------------------------------------- trajectory frames
// A ring of pinned host slots. A capture notes the slot and the state buffers of the moment; the kernels that unpack (or
// round) the state are launched by the next step on its bonded stream, behind the bonded kernel and ahead of the marker the
// update waits for, and write straight into the slot's pinned memory. On the main stream a capture costs nothing, and no
// copy follows: a device-to-host copy is a kernel of the runtime's (30 us for 1.7 MB, on compute units, beside the energy
// step's constraint kernel), and in front of it the staging kernel and its hand-over held the main stream for 12 us per frame.
// What a capture notes stays as it is until the next step's update: nothing ahead of the join touches positions, velocities
// or the box, and the update (which writes the other position buffer, the velocities in place and, with pressure coupling,
// the box) waits for the marker behind the capture's kernels. A waiting thread that launches them itself does so under the
// ring's lock on the main stream, i.e. before the stepping thread can have enqueued that update. Anything else that
// rewrites the state launches the pending captures first (mdx_frames_flush). A stream of their own for the frames would be the fourth of the
// process: with four hardware queues per process it shares a queue with a compute stream and stalls it (measured twice:
// +130 us per step in an early form of the step, 170 against 162 us per step in the present one).
// A writer thread waits for the slot's event and encodes the frame.
__global__ __launch_bounds__(256) void k_unpack3(int n, const float4 *__restrict__ a, float *__restrict__ out,
                                                 const double *__restrict__ boxd, double *__restrict__ box_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 9 && box_out) box_out[i] = boxd[i];
    if (i >= n) return;
    const float4 p = a[i];
    out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
}

// Capture for the compressed trajectory: the rounding pass of the xtc coordinate block on the GPU. Per atom
// l = (int)(x * precision +- 0.5) in single precision (separately rounded multiply and add, as the host encoder computes
// it), per frame the extent of the integers, the smallest sum of absolute differences between neighbouring atoms (it
// selects the first entry of the format's small-number table) and a flag for a coordinate that is not finite or does
// not fit; the host is left with the bit packing (csrc/trajio.cpp, trj_xtc_encode_frame_q).
// stats: min x y z, max x y z, smallest neighbour difference, flag (set up by k_frame_stats_init).
__device__ __forceinline__ bool xtc_round(float xf, float precision, int &l)
{
    const float lf = __fmul_rn(xf, precision);
    if (!(fabsf(lf) <= (float)(INT_MAX - 2))) { l = 0; return false; }       // also catches NaN
    l = (int)(lf >= 0.0f ? __fadd_rn(lf, 0.5f) : __fsub_rn(lf, 0.5f));
    return true;
}

// The frame's statistics (minima 0-2, maxima 3-5, smallest difference between consecutive atoms 6, a coordinate out of
// range 7) leave the kernel as one set of eight per workgroup, written with plain stores; the host folds the
// MDX_FRAME_WGS sets when it takes the frame. (One set updated with atomics needed a launch to initialise it, and the
// 543 workgroups of a thread-per-atom launch queued up at its eight addresses: 17 us for 139 k atoms against 4.)
#define MDX_FRAME_WGS 128
__global__ __launch_bounds__(256) void k_frame_quantize(int n, const float4 *__restrict__ a, float precision, int *__restrict__ out,
                                                        int *__restrict__ stats, const double *__restrict__ boxd,
                                                        double *__restrict__ box_out)
{
    __shared__ int red[8][4];
    if (blockIdx.x == 0 && threadIdx.x < 9 && box_out) box_out[threadIdx.x] = boxd[threadIdx.x];
    int mn[3] = { INT_MAX, INT_MAX, INT_MAX }, mx[3] = { INT_MIN, INT_MIN, INT_MIN }, md = INT_MAX, bad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = a[i];
        int l[3];
        const bool ok = xtc_round(p.x, precision, l[0]) & xtc_round(p.y, precision, l[1]) & xtc_round(p.z, precision, l[2]);
        if (!ok) { bad = 1; continue; }
        for (int d = 0; d < 3; d++) { mn[d] = min(mn[d], l[d]); mx[d] = max(mx[d], l[d]); }
        out[3 * (size_t)i] = l[0]; out[3 * (size_t)i + 1] = l[1]; out[3 * (size_t)i + 2] = l[2];
        if (i > 0) {
            const float4 q = a[i - 1];
            int o[3];
            if (xtc_round(q.x, precision, o[0]) & xtc_round(q.y, precision, o[1]) & xtc_round(q.z, precision, o[2])) {
                const long long diff = llabs((long long)o[0] - l[0]) + llabs((long long)o[1] - l[1]) + llabs((long long)o[2] - l[2]);
                md = min(md, (int)(diff < (long long)INT_MAX ? diff : (long long)INT_MAX));
            }
        }
    }
    // eight values per thread -> per wave (shuffles) -> per workgroup (LDS) -> the workgroup's set
    int v[8] = { mn[0], mn[1], mn[2], mx[0], mx[1], mx[2], md, bad };
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        for (int o = 32; o; o >>= 1) {
            const int w = __shfl_xor(v[k], o, 64);
            v[k] = (k >= 3 && k < 6) || k == 7 ? max(v[k], w) : min(v[k], w);
        }
        if (lane == 0) red[k][wv] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        const bool is_max = (k >= 3 && k < 6) || k == 7;
        int r = red[k][0];
        for (int w = 1; w < 4; w++) r = is_max ? max(r, red[k][w]) : min(r, red[k][w]);
        stats[8 * blockIdx.x + k] = r;
    }
}

struct FrameSlot { float *x = nullptr, *v = nullptr; double *box = nullptr; int *stats = nullptr; hipEvent_t ev = nullptr; long long step = -1; int has_v = 0; int quantized = 0;
                   bool enqueued = true; const float4 *src_x = nullptr, *src_v = nullptr; float precision = 0.0f; };
// A slot's positions (or xtc integers), statistics and box are one pinned block:
// [x: 3 n words | pad to 8 bytes | stats: 8 MDX_FRAME_WGS ints | box: 9 doubles].
// `pending`: slots captured whose kernels are not launched yet; `lock` guards it and the slots' `enqueued` flags (the stepping
// thread launches them, a writer thread waiting for a frame looks and, if no step comes, launches them itself).
struct FrameRing { std::vector<FrameSlot> slots; size_t x_words = 0, slot_words = 0; std::vector<int> pending; std::mutex lock; };
static FrameRing *ring_of(mdx_system *s) { return (FrameRing *)s->frame_ring; }

void mdx_frames_free(mdx_system *s)
{
    FrameRing *r = ring_of(s);
    if (!r) return;
    for (FrameSlot &f : r->slots) {
        if (f.x) (void)hipHostFree(f.x);          // (statistics and box are part of this block)
        if (f.v) (void)hipHostFree(f.v);
        if (f.ev) (void)hipEventDestroy(f.ev);
    }
    delete r;
    s->frame_ring = nullptr;
}

extern "C" int mdx_frame_ring(mdx_system *s, int nslots)
{
    (void)hipSetDevice(s->device);
    ARG_CHECK(s, nslots >= 1 && nslots <= 64, "mdx_frame_ring: 1..64 slots");
    mdx_frames_free(s);
    FrameRing *r = new FrameRing();
    s->frame_ring = r;
    r->slots.resize(nslots);
    const size_t cnt = 3 * (size_t)s->n;
    r->x_words = (cnt + 1) & ~(size_t)1;                                   // statistics and box start on an 8-byte boundary
    r->slot_words = r->x_words + 8 * MDX_FRAME_WGS + 18;
    for (FrameSlot &f : r->slots) {
        MDX_CHECK(s, hipHostMalloc((void **)&f.x, sizeof(float) * r->slot_words, hipHostMallocDefault));
        MDX_CHECK(s, hipHostMalloc((void **)&f.v, sizeof(float) * cnt, hipHostMallocDefault));
        f.stats = (int *)(f.x + r->x_words);
        f.box = (double *)(f.stats + 8 * MDX_FRAME_WGS);
        MDX_CHECK(s, hipEventCreateWithFlags(&f.ev, hipEventDisableTiming));
    }
    return MDX_OK;
}

static int flush_locked(mdx_system *s, FrameRing *r, hipStream_t st)
{
    const int n = s->n, nb = (n + 255) / 256;
    for (int slot : r->pending) {
        FrameSlot &f = r->slots[slot];
        if (f.has_v) hipLaunchKernelGGL(k_unpack3, dim3(nb), dim3(256), 0, st, n, f.src_v, f.v, nullptr, nullptr);
        if (f.quantized)
            hipLaunchKernelGGL(k_frame_quantize, dim3(MDX_FRAME_WGS), dim3(256), 0, st, n, f.src_x, f.precision, (int *)f.x, f.stats,
                               (const double *)s->d->boxd, f.box);
        else
            hipLaunchKernelGGL(k_unpack3, dim3(nb), dim3(256), 0, st, n, f.src_x, f.x, (const double *)s->d->boxd, f.box);
        MDX_CHECK(s, hipGetLastError());
        MDX_CHECK(s, hipEventRecord(f.ev, st));
        f.enqueued = true;
    }
    r->pending.clear();
    return MDX_OK;
}
// Launch the kernels of the frames captured since the last call on stream `st`, behind whatever it holds.
int mdx_frames_flush(mdx_system *s, hipStream_t st, bool *any)
{
    FrameRing *r = ring_of(s);
    if (!r) return MDX_OK;
    std::lock_guard<std::mutex> g(r->lock);
    if (r->pending.empty()) return MDX_OK;
    if (any) *any = true;
    return flush_locked(s, r, st);
}

static int capture_frame(mdx_system *s, int slot, int want_v, bool quantize, float precision)
{
    (void)hipSetDevice(s->device);
    FrameRing *r = ring_of(s);
    ARG_CHECK(s, r && slot >= 0 && slot < (int)r->slots.size(), "mdx_capture_frame: call mdx_frame_ring first / bad slot");
    FrameSlot &f = r->slots[slot];
    std::lock_guard<std::mutex> g(r->lock);
    ARG_CHECK(s, f.enqueued, "mdx_capture_frame: the slot's last capture has not been launched yet");
    f.step = s->step; f.has_v = want_v ? 1 : 0; f.quantized = quantize ? 1 : 0; f.precision = precision;
    f.src_x = (const float4 *)s->x; f.src_v = (const float4 *)s->v; f.enqueued = false;
    r->pending.push_back(slot);
    // (a decomposed run exchanges halo positions into the state between steps: no deferral there)
    if (s->dd_on) return flush_locked(s, r, s->stream);
    return MDX_OK;
}

extern "C" int mdx_capture_frame(mdx_system *s, int slot, int want_v) { return capture_frame(s, slot, want_v, false, 0.0f); }

extern "C" int mdx_capture_frame_xtc(mdx_system *s, int slot, int want_v, float precision)
{
    ARG_CHECK(s, precision > 0.0f, "mdx_capture_frame_xtc: precision must be positive");
    return capture_frame(s, slot, want_v, true, precision);
}

// A waiting thread gives the stepping thread two milliseconds to reach the point of its step where it launches the capture
// (it gets there within a step's enqueue time), then launches it itself on the main stream: a caller that captures and waits
// without stepping.
static int await_launched(mdx_system *s, FrameRing *r, FrameSlot &f)
{
    std::unique_lock<std::mutex> g(r->lock);
    for (int spin = 0; !f.enqueued && spin < 40; spin++) {
        g.unlock();
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        g.lock();
    }
    return f.enqueued ? MDX_OK : flush_locked(s, r, s->stream);
}

// Blocks until the capture into `slot` has landed; the pointers stay valid until the slot is
// captured into again. May be called from another thread than the one that steps.
extern "C" int mdx_frame_wait(mdx_system *s, int slot, const float **x, const float **v, double *box9, long long *step)
{
    FrameRing *r = ring_of(s);
    ARG_CHECK(s, r && slot >= 0 && slot < (int)r->slots.size(), "mdx_frame_wait: bad slot");
    FrameSlot &f = r->slots[slot];
    ARG_CHECK(s, f.step >= 0, "mdx_frame_wait: nothing captured in this slot");
    ARG_CHECK(s, !f.quantized, "mdx_frame_wait: the slot holds xtc integers (mdx_frame_wait_xtc)");
    (void)hipSetDevice(s->device);
    if (int rc = await_launched(s, r, f)) return rc;
    if (hipEventSynchronize(f.ev) != hipSuccess) { return MDX_ERR_HIP; }
    if (x) *x = f.x;
    if (v) *v = f.has_v ? f.v : nullptr;
    if (box9) memcpy(box9, f.box, sizeof(double) * 9);
    if (step) *step = f.step;
    return MDX_OK;
}

// The same for a slot captured with mdx_capture_frame_xtc: integer coordinates and the frame's eight statistics.
extern "C" int mdx_frame_wait_xtc(mdx_system *s, int slot, const int **ix, const float **v, double *box9, long long *step, int *stats8)
{
    FrameRing *r = ring_of(s);
    ARG_CHECK(s, r && slot >= 0 && slot < (int)r->slots.size(), "mdx_frame_wait_xtc: bad slot");
    FrameSlot &f = r->slots[slot];
    ARG_CHECK(s, f.step >= 0 && f.quantized, "mdx_frame_wait_xtc: no xtc capture in this slot");
    (void)hipSetDevice(s->device);
    if (int rc = await_launched(s, r, f)) return rc;
    if (hipEventSynchronize(f.ev) != hipSuccess) { return MDX_ERR_HIP; }
    if (ix) *ix = (const int *)f.x;
    if (v) *v = f.has_v ? f.v : nullptr;
    if (box9) memcpy(box9, f.box, sizeof(double) * 9);
    if (step) *step = f.step;
    if (stats8) {
        // fold the workgroups' sets: minima, maxima, the smallest difference, the out-of-range flag
        for (int k = 0; k < 8; k++) {
            const bool is_max = (k >= 3 && k < 6) || k == 7;
            int r = f.stats[k];
            for (int w = 1; w < MDX_FRAME_WGS; w++) r = is_max ? std::max(r, f.stats[8 * w + k]) : std::min(r, f.stats[8 * w + k]);
            stats8[k] = r;
        }
    }
    return MDX_OK;
}
