// Device code of the centre-of-mass pull: included by pull.hip, whose kernel runs it on the bonded stream, and by pme.hip,
// whose spreading launch carries it along on the steps without energies.
#pragma once
#include "mdx_internal.h"

struct PullArgs {
    int n1, n2;
    const int *g1, *g2;
    int pbc1, pbc2;
    int type;
    double k, rate, init_user;
    int start;
    int dim[3];
    double time, dt;
    double *log; int log_cap; long long log_slot; int log_now;
    int energy;
};

// The pull of one step, by one workgroup of 256 threads (the kernel below, or a workgroup of the charge-spreading launch on
// the steps where the bonded terms ride there too).
__device__ __forceinline__ void pull_block(const PullArgs &P, const float4 *__restrict__ x, const float4 *__restrict__ v,
                                           float *__restrict__ f, DevState *dw)
{
    __shared__ double red[16];
    __shared__ double com[2][4];
    __shared__ double sh_dr[4];
    __shared__ double sh_f;
    const DevBox box = dw->box;
    for (int g = 0; g < 2; g++) {
        const int n = g == 0 ? P.n1 : P.n2;
        const int *idx = g == 0 ? P.g1 : P.g2;
        const int pa = g == 0 ? P.pbc1 : P.pbc2;
        const float3 xr = xyz(x[idx[pa]]);
        double s[4] = { 0, 0, 0, 0 };
        for (int t = threadIdx.x; t < n; t += blockDim.x) {
            const int a = idx[t];
            const float3 d = mdx_min_image(box, xyz(x[a]) - xr);
            const double m = 1.0 / (double)v[a].w;
            s[0] += m * d.x; s[1] += m * d.y; s[2] += m * d.z; s[3] += m;
        }
        for (int k = 0; k < 4; k++) {
            const double r = block_sum_d(s[k], red);
            if (threadIdx.x == 0) com[g][k] = r;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double im = 1.0 / com[g][3];
            com[g][0] = xr.x + com[g][0] * im; com[g][1] = xr.y + com[g][1] * im; com[g][2] = xr.z + com[g][2] * im;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float3 d = make_float3((float)(com[1][0] - com[0][0]), (float)(com[1][1] - com[0][1]), (float)(com[1][2] - com[0][2]));
        d = mdx_min_image(box, d);
        double dr[3] = { P.dim[0] ? d.x : 0.0, P.dim[1] ? d.y : 0.0, P.dim[2] ? d.z : 0.0 };
        const double r = sqrt(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
        if (!dw->pull_have_init) {
            dw->pull_init = P.init_user + (P.start ? r : 0.0);
            dw->pull_have_init = 1;
            dw->pull_value = r;
            dw->pull_work = 0.0;
        }
        double fs, epot, ref = 0.0;
        if (P.type == 0) {
            fs = -P.k;
            epot = P.k * r;
            dw->pull_work += fs * (r - dw->pull_value);
        } else {
            ref = dw->pull_init + P.rate * P.time;
            fs = -P.k * (r - ref);
            epot = 0.5 * P.k * (r - ref) * (r - ref);
            dw->pull_work += fs * P.rate * P.dt;
        }
        dw->pull_value = r; dw->pull_force = fs; dw->pull_ref = ref;
        if (P.energy) dw->ener[MDX_E_PULL] += epot;
        const double ir = r > 0 ? 1.0 / r : 0.0;
        sh_dr[0] = dr[0] * ir; sh_dr[1] = dr[1] * ir; sh_dr[2] = dr[2] * ir; sh_dr[3] = r;
        sh_f = fs;
        if (P.energy) {
            // virial of the pair of COM forces: -1/2 dr (x) F2
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) dw->vir[3 * a + b] += -0.5 * dr[a] * (fs * dr[b] * ir);
        }
        if (P.log_now && P.log) {
            double *rec = P.log + 4 * (P.log_slot % P.log_cap);
            rec[0] = P.time; rec[1] = r; rec[2] = fs; rec[3] = dw->pull_work;
            __threadfence_system();
        }
    }
    __syncthreads();
    const double fs = sh_f;
    for (int g = 0; g < 2; g++) {
        const int n = g == 0 ? P.n1 : P.n2;
        const int *idx = g == 0 ? P.g1 : P.g2;
        const double sgn = g == 0 ? -1.0 : 1.0;
        const double scale = sgn * fs / com[g][3];
        for (int t = threadIdx.x; t < n; t += blockDim.x) {
            const int a = idx[t];
            const double m = 1.0 / (double)v[a].w;
            atomicAdd(&f[3 * a + 0], (float)(scale * m * sh_dr[0]));
            atomicAdd(&f[3 * a + 1], (float)(scale * m * sh_dr[1]));
            atomicAdd(&f[3 * a + 2], (float)(scale * m * sh_dr[2]));
        }
    }
}

