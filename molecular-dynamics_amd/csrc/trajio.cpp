// Trajectory file codecs on the host: compressed `.xtc` frames and full-precision `.trr` frames,
// both XDR (big-endian) streams.
// Written for the output cadence the pipeline's .mdp files ask for: `nstxout-compressed`
// (/root/reference/mdp/production.mdp:12, npt_complex.mdp:17, pull_short.mdp:8) and
// `nstxout/nstvout` (/root/reference/mdp/nvt_complex.mdp:8-9, production.mdp:7); the files are
// what `gmx rms`, `gmx trjconv` and gmx_MMPBSA read afterwards
// (/root/reference/Snakefile:348,1054-1055).
//
// The .xtc coordinate block follows the published xtc3 scheme: coordinates are multiplied by the
// precision and rounded to integers; each atom is sent either as a "large" triple relative to the
// frame minimum (mixed-radix packed into the fewest bits that hold the frame's extent) or, for
// runs of up to eight atoms that lie close to their predecessor (water molecules), as "small"
// triples relative to the previous atom, packed with a radix from a table of ~2^(k/3) sizes that
// adapts up or down by one entry per run. The first two atoms of a run are interchanged so that
// a water's oxygen-hydrogen differences stay small.
//
// This library has no GPU code: it is built with the host compiler and is usable on its own.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <math.h>
#include <time.h>
#include <vector>
#include <atomic>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>

extern "C" {

enum { TRJ_OK = 0, TRJ_ERR_IO = -1, TRJ_ERR_RANGE = -2, TRJ_ERR_FORMAT = -3, TRJ_EOF = 1 };

static const int kMagic[] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 10, 12, 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024,
    1290, 1625, 2048, 2580, 3250, 4096, 5060, 6501, 8192, 10321, 13003, 16384, 20642, 26007, 32768, 41285, 52015, 65536, 82570,
    104031, 131072, 165140, 208063, 262144, 330280, 416127, 524287, 660561, 832255, 1048576, 1321122, 1664510, 2097152, 2642245,
    3329021, 4194304, 5284491, 6658042, 8388607, 10568983, 13316085, 16777216
};
static const int kFirstIdx = 9;
static const int kLastIdx = (int)(sizeof(kMagic) / sizeof(kMagic[0]));
static const float kMaxAbs = (float)(INT_MAX - 2);

// ---------------------------------------------------------------- bit stream (MSB first)
// (aligned to two cache lines: the writers of a frame's segments sit side by side in one array and every put updates
// len / acc / nacc; sharing a line made three or more segment threads slower than one thread for the whole frame)
struct alignas(128) BitWriter {
    std::vector<unsigned char> buf;     // sized up front by the caller for the worst case (no growth checks in put)
    size_t len = 0;
    uint64_t acc = 0;                   // low `nacc` bits are pending, nacc < 8 between calls
    int nacc = 0;
    void reserve_bytes(size_t n) { if (buf.size() < n) buf.resize(n); }
    // append the low `nbits` (<= 32) bits of v, most significant first
    inline void put(int nbits, unsigned int v)
    {
        if (nbits <= 0) return;
        const uint64_t mask = nbits >= 32 ? 0xFFFFFFFFull : ((1ull << nbits) - 1ull);
        acc = (acc << nbits) | ((uint64_t)v & mask);
        nacc += nbits;
        while (nacc >= 8) { buf[len++] = (unsigned char)(acc >> (nacc - 8)); nacc -= 8; }
        acc &= (1ull << nacc) - 1ull;
    }
    // append the bit string of another writer, its pending bits included (the other writer stays usable)
    void append(const BitWriter &o)
    {
        if (buf.size() < len + o.len + 16) buf.resize(len + o.len + 16);
        if (nacc == 0) {
            memcpy(buf.data() + len, o.buf.data(), o.len);
            len += o.len;
        } else {
            // the source bytes shifted right by the nacc pending bits, eight at a time
            const int sh = nacc;
            uint64_t carry = acc;                       // low `sh` bits pending
            size_t k = 0;
            unsigned char *dst = buf.data() + len;
            for (; k + 8 <= o.len; k += 8) {
                uint64_t v; memcpy(&v, o.buf.data() + k, 8); v = __builtin_bswap64(v);
                const uint64_t out = (carry << (64 - sh)) | (v >> sh);
                const uint64_t be = __builtin_bswap64(out);
                memcpy(dst + k, &be, 8);
                carry = v & ((1ull << sh) - 1ull);
            }
            len += k; acc = carry;
            for (; k < o.len; k++) put(8, o.buf[k]);
        }
        if (o.nacc > 0) put(o.nacc, (unsigned)o.acc);
    }
    size_t finish()
    {
        if (nacc > 0) {
            if (len + 1 > buf.size()) buf.resize(len + 1);
            buf[len++] = (unsigned char)((acc << (8 - nacc)) & 0xFFu); nacc = 0; acc = 0;
        }
        return len;
    }
};

struct BitReader {
    const unsigned char *p; size_t n, pos = 0; unsigned int acc = 0; int nacc = 0; bool bad = false;
    BitReader(const unsigned char *d, size_t len) : p(d), n(len) {}
    unsigned int get(int nbits)
    {
        unsigned int out = 0;
        while (nbits > 0) {
            if (nacc == 0) { if (pos >= n) { bad = true; return 0; } acc = p[pos++]; nacc = 8; }
            const int take = nbits < nacc ? nbits : nacc;
            out = (out << take) | ((acc >> (nacc - take)) & ((1u << take) - 1u));
            nacc -= take; nbits -= take;
        }
        return out;
    }
};

static int bits_of(unsigned int size)
{
    unsigned int num = 1; int nbits = 0;
    while (size >= num && nbits < 32) { nbits++; num <<= 1; }
    return nbits;
}

// bits needed for a mixed-radix number with the given radices
static int bits_of_product(const unsigned int sizes[3])
{
    unsigned __int128 prod = 1;
    for (int i = 0; i < 3; i++) prod *= sizes[i];
    // value of the highest byte decides the bits above the whole bytes
    int nbytes = 0; unsigned __int128 t = prod;
    unsigned int top = 0;
    do { top = (unsigned int)(t & 0xFF); t >>= 8; nbytes++; } while (t != 0);
    int nbits = 0; unsigned int num = 1;
    while (top >= num) { nbits++; num *= 2; }
    return nbits + (nbytes - 1) * 8;
}

// three numbers, radices sizes[1], sizes[2]: value = (n0*s1 + n1)*s2 + n2, sent low byte first
static void put_triple(BitWriter &w, int nbits, const unsigned int sizes[3], const unsigned int nums[3])
{
    if (nbits <= 64) {
        // the value fits a machine word: whole bytes low byte first, then the remaining high bits
        const uint64_t v = ((uint64_t)nums[0] * sizes[1] + nums[1]) * sizes[2] + nums[2];
        const int k = nbits >> 3, r = nbits & 7;
        // as one big-endian bit string: the k low bytes reversed, followed by the r remaining high bits
        uint64_t word = k ? (__builtin_bswap64(v) >> (64 - 8 * k)) : 0;
        if (r) word = (k == 8 ? 0 : (word << r)) | (k == 8 ? 0 : (v >> (8 * k)));
        if (nbits > 32) { w.put(nbits - 32, (unsigned int)(word >> 32)); w.put(32, (unsigned int)word); }
        else w.put(nbits, (unsigned int)word);
        return;
    }
    unsigned __int128 v = nums[0];
    v = v * sizes[1] + nums[1];
    v = v * sizes[2] + nums[2];
    unsigned char bytes[16]; int nbytes = 0;
    do { bytes[nbytes++] = (unsigned char)(v & 0xFF); v >>= 8; } while (v != 0);
    if (nbits >= nbytes * 8) {
        for (int i = 0; i < nbytes; i++) w.put(8, bytes[i]);
        w.put(nbits - nbytes * 8, 0);
    } else {
        int i = 0;
        for (; i < nbytes - 1; i++) w.put(8, bytes[i]);
        w.put(nbits - (nbytes - 1) * 8, bytes[i]);
    }
}

static void get_triple(BitReader &r, int nbits, const unsigned int sizes[3], int out[3])
{
    unsigned __int128 v = 0; int shift = 0;
    while (nbits > 8) { v |= (unsigned __int128)r.get(8) << shift; shift += 8; nbits -= 8; }
    if (nbits > 0) v |= (unsigned __int128)r.get(nbits) << shift;
    out[2] = (int)(v % sizes[2]); v /= sizes[2];
    out[1] = (int)(v % sizes[1]); v /= sizes[1];
    out[0] = (int)v;
}

// ---------------------------------------------------------------- XDR primitives
static void xdr_put_u32(std::vector<unsigned char> &o, uint32_t v)
{
    o.push_back((unsigned char)(v >> 24)); o.push_back((unsigned char)(v >> 16)); o.push_back((unsigned char)(v >> 8)); o.push_back((unsigned char)v);
}
static void xdr_put_f32(std::vector<unsigned char> &o, float f) { uint32_t u; memcpy(&u, &f, 4); xdr_put_u32(o, u); }
static uint32_t be32(const unsigned char *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static float bef32(const unsigned char *p) { uint32_t u = be32(p); float f; memcpy(&f, &u, 4); return f; }

}  // extern "C" (templates have C++ linkage)
// fn(k) for k = 0 .. T-1, k = 0 on the calling thread and the others on workers that belong to the calling thread and
// sleep on a condition variable between calls (they never spin next to the step loop). Starting threads per call cost
// 30-40 us each, one after the other: at a dozen slices and two calls per frame that was most of a millisecond of a
// frame that is otherwise compressed in half of one.
class SlicePool {
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::vector<std::thread> th;
    const std::function<void(int)> *job = nullptr;
    int T = 0, left = 0;
    unsigned long long gen = 0;
    bool stop = false;
    void worker(int k)
    {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)> *j = nullptr;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                if (k < T) j = job;
            }
            if (!j) continue;
            (*j)(k);
            std::lock_guard<std::mutex> l(m);
            if (--left == 0) cv_done.notify_one();
        }
    }
public:
    void run(int nslices, const std::function<void(int)> &fn)
    {
        if (nslices <= 1) { fn(0); return; }
        while ((int)th.size() < nslices - 1) { const int k = (int)th.size() + 1; th.emplace_back([this, k] { worker(k); }); }
        {
            std::lock_guard<std::mutex> l(m);
            job = &fn; T = nslices; left = nslices - 1; gen++;
        }
        cv.notify_all();
        fn(0);
        std::unique_lock<std::mutex> l(m);
        cv_done.wait(l, [&] { return left == 0; });
        job = nullptr;
    }
    ~SlicePool()
    {
        { std::lock_guard<std::mutex> l(m); stop = true; }
        cv.notify_all();
        for (std::thread &t : th) t.join();
    }
};
// One pool per calling thread, shared by every call site (a thread_local inside the template below would be one pool
// per instantiation, i.e. per lambda).
static SlicePool &thread_slice_pool()
{
    static thread_local SlicePool pool;
    return pool;
}
template <typename F>
static void run_slices(int T, F fn)
{
    if (T <= 1) { fn(0); return; }
    const std::function<void(int)> f = [&fn](int k) { fn(k); };
    thread_slice_pool().run(T, f);
}
extern "C" {

// segments of multi-threaded frames that had to be encoded again from the true state (diagnostic counter)
static std::atomic<long long> g_xtc_respeculated{0};
static thread_local double g_xtc_phase[4 + 16] = { 0 };   // last frame packed by this thread: segments, merge, copy-out, workers' longest (ms)
static double now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static int g_xtc_warm = 2048;      // atoms a segment's thread walks through, without output, ahead of its segment

// ---------------------------------------------------------------- xtc run encoder
// Frame-wide constants of the coordinate block.
struct XtcParams {
    int minint[3];
    unsigned int sizeint[3], bitsizeint[3];
    int bitsize;
    int minidx, maxidx, larger;
};
// What the encoder carries from one run of atoms to the next.
struct XtcState {
    int i;              // next atom
    int smallidx;       // current entry of the small-number table
    int smallnum, smaller;   // half the table entry, and half the entry below (history dependent at the table's lower end)
    int prevrun;        // length code of the last run header written
    int prev[3];        // last atom sent
    bool operator==(const XtcState &o) const
    {
        return i == o.i && smallidx == o.smallidx && smallnum == o.smallnum && smaller == o.smaller && prevrun == o.prevrun &&
               prev[0] == o.prev[0] && prev[1] == o.prev[1] && prev[2] == o.prev[2];
    }
};

// Encode whole runs starting at atom st.i until st.i >= stop (a run is never cut, so st.i may end beyond stop).
// With w == nullptr only the state is advanced. The integer coordinates are not modified: the interchange of the
// first two atoms of a run is done on the fly.
static void xtc_encode_range(const int *ip, int natoms, const XtcParams &P, XtcState &st, int stop, BitWriter *w)
{
    int i = st.i, smallidx = st.smallidx, prevrun = st.prevrun;
    int prev[3] = { st.prev[0], st.prev[1], st.prev[2] };
    int smaller = st.smaller, smallnum = st.smallnum;
    unsigned int sizesmall[3] = { (unsigned)kMagic[smallidx], (unsigned)kMagic[smallidx], (unsigned)kMagic[smallidx] };
    unsigned int tmp[30];
    while (i < stop && i < natoms) {
        const int *c = ip + 3 * (size_t)i;
        bool is_small = false;
        int is_smaller;
        if (smallidx < P.maxidx && i >= 1 && abs(c[0] - prev[0]) < P.larger && abs(c[1] - prev[1]) < P.larger && abs(c[2] - prev[2]) < P.larger)
            is_smaller = 1;
        else if (smallidx > P.minidx)
            is_smaller = -1;
        else
            is_smaller = 0;
        // a second atom close to the first: the two are interchanged (the second is sent in full, the first as the
        // run's first small difference), which keeps a water's oxygen-hydrogen differences small
        const int *first = c, *second = nullptr;
        if (i + 1 < natoms && abs(c[0] - c[3]) < smallnum && abs(c[1] - c[4]) < smallnum && abs(c[2] - c[5]) < smallnum) {
            first = c + 3; second = c;
            is_small = true;
        }
        if (w) {
            unsigned int big[3] = { (unsigned)(first[0] - P.minint[0]), (unsigned)(first[1] - P.minint[1]), (unsigned)(first[2] - P.minint[2]) };
            if (P.bitsize == 0) { w->put(P.bitsizeint[0], big[0]); w->put(P.bitsizeint[1], big[1]); w->put(P.bitsizeint[2], big[2]); }
            else put_triple(*w, P.bitsize, P.sizeint, big);
        }
        prev[0] = first[0]; prev[1] = first[1]; prev[2] = first[2];
        i++;
        int run = 0;
        if (!is_small && is_smaller == -1) is_smaller = 0;
        while (is_small && run < 8 * 3) {
            const int *cur = second ? second : ip + 3 * (size_t)i;
            second = nullptr;
            if (is_smaller == -1) {
                const long long dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
                if (dx * dx + dy * dy + dz * dz >= (long long)smaller * smaller) is_smaller = 0;
            }
            tmp[run++] = (unsigned)(cur[0] - prev[0] + smallnum);
            tmp[run++] = (unsigned)(cur[1] - prev[1] + smallnum);
            tmp[run++] = (unsigned)(cur[2] - prev[2] + smallnum);
            prev[0] = cur[0]; prev[1] = cur[1]; prev[2] = cur[2];
            i++;
            is_small = false;
            if (i < natoms) {
                const int *nx = ip + 3 * (size_t)i;
                if (abs(nx[0] - prev[0]) < smallnum && abs(nx[1] - prev[1]) < smallnum && abs(nx[2] - prev[2]) < smallnum) is_small = true;
            }
        }
        if (run != prevrun || is_smaller != 0) {
            prevrun = run;
            if (w) { w->put(1, 1); w->put(5, (unsigned)(run + is_smaller + 1)); }
        } else if (w) {
            w->put(1, 0);
        }
        if (w) for (int k = 0; k < run; k += 3) put_triple(*w, smallidx, sizesmall, &tmp[k]);
        if (is_smaller != 0) {
            smallidx += is_smaller;
            if (is_smaller < 0) { smallnum = smaller; smaller = kMagic[smallidx - 1] / 2; }
            else { smaller = smallnum; smallnum = kMagic[smallidx] / 2; }
            sizesmall[0] = sizesmall[1] = sizesmall[2] = (unsigned)kMagic[smallidx];
        }
    }
    st.i = i; st.smallidx = smallidx; st.smallnum = smallnum; st.smaller = smaller; st.prevrun = prevrun;
    st.prev[0] = prev[0]; st.prev[1] = prev[1]; st.prev[2] = prev[2];
}

// ---------------------------------------------------------------- xtc coordinate block
static int xtc_pack_coords(std::vector<unsigned char> &o, int natoms, const int *ip, const int *minint, const int *maxint, int mindiff,
                           int nthreads);

// Appends `natoms` (int), and the raw or compressed coordinates to `o`.
static int xtc_encode_coords(std::vector<unsigned char> &o, int natoms, const float *x, float precision, int nthreads)
{
    xdr_put_u32(o, (uint32_t)natoms);
    if (natoms <= 9) {
        for (int i = 0; i < 3 * natoms; i++) xdr_put_f32(o, x[i]);
        return TRJ_OK;
    }
    if (precision <= 0) precision = 1000.0f;
    xdr_put_f32(o, precision);
    // per-thread scratch that only grows: no per-frame allocation or zero fill
    static thread_local std::vector<int> ip_scratch;
    if (ip_scratch.size() < 3 * (size_t)natoms + 3) ip_scratch.resize(3 * (size_t)natoms + 3);
    int *const ip = ip_scratch.data();
    int minint[3] = { INT_MAX, INT_MAX, INT_MAX }, maxint[3] = { INT_MIN, INT_MIN, INT_MIN };
    int mindiff = INT_MAX;
    {
        // rounding to integers, extent and the smallest neighbour distance: independent slices of atoms
        const int TQ = (nthreads > 1 && natoms >= 8192) ? (nthreads > 16 ? 16 : nthreads) : 1;
        int mn[16][3], mx[16][3], md[16], bad[16];
        run_slices(TQ, [&](int k) {
            const int b0 = (int)((long long)natoms * k / TQ), b1 = (int)((long long)natoms * (k + 1) / TQ);
            int lmn[3] = { INT_MAX, INT_MAX, INT_MAX }, lmx[3] = { INT_MIN, INT_MIN, INT_MIN }, lmd = INT_MAX, lbad = 0;
            int old[3] = { 0, 0, 0 };
            if (b0 > 0)
                for (int d = 0; d < 3; d++) { const float lf = x[3 * (size_t)(b0 - 1) + d] * precision; old[d] = (int)(lf >= 0 ? lf + 0.5f : lf - 0.5f); }
            for (int i = b0; i < b1; i++) {
                int l[3];
                for (int d = 0; d < 3; d++) {
                    const float lf = x[3 * (size_t)i + d] * precision;
                    if (!(fabsf(lf) <= kMaxAbs)) { lbad = 1; l[d] = 0; continue; }       // also catches NaN
                    l[d] = (int)(lf >= 0 ? lf + 0.5f : lf - 0.5f);
                    if (l[d] < lmn[d]) lmn[d] = l[d];
                    if (l[d] > lmx[d]) lmx[d] = l[d];
                    ip[3 * (size_t)i + d] = l[d];
                }
                const long long diff = llabs((long long)old[0] - l[0]) + llabs((long long)old[1] - l[1]) + llabs((long long)old[2] - l[2]);
                if (i > 0 && diff < lmd) lmd = (int)diff;
                old[0] = l[0]; old[1] = l[1]; old[2] = l[2];
            }
            for (int d = 0; d < 3; d++) { mn[k][d] = lmn[d]; mx[k][d] = lmx[d]; }
            md[k] = lmd; bad[k] = lbad;
        });
        for (int k = 0; k < TQ; k++) {
            if (bad[k]) return TRJ_ERR_RANGE;
            for (int d = 0; d < 3; d++) { if (mn[k][d] < minint[d]) minint[d] = mn[k][d]; if (mx[k][d] > maxint[d]) maxint[d] = mx[k][d]; }
            if (md[k] < mindiff) mindiff = md[k];
        }
    }
    return xtc_pack_coords(o, natoms, ip, minint, maxint, mindiff, nthreads);
}

// The part of the coordinate block behind the precision: extents, first small-number index, and the bit stream of the
// integer coordinates `ip` (3 per atom, already rounded; extents and smallest neighbour distance as the rounding pass
// found them, on the host above or on the GPU).
static int xtc_pack_coords(std::vector<unsigned char> &o, int natoms, const int *ip, const int *minint, const int *maxint, int mindiff,
                           int nthreads)
{
    for (int d = 0; d < 3; d++) xdr_put_u32(o, (uint32_t)minint[d]);
    for (int d = 0; d < 3; d++) xdr_put_u32(o, (uint32_t)maxint[d]);
    for (int d = 0; d < 3; d++) if ((float)maxint[d] - (float)minint[d] >= kMaxAbs) return TRJ_ERR_RANGE;
    unsigned int sizeint[3], bitsizeint[3] = { 0, 0, 0 };
    for (int d = 0; d < 3; d++) sizeint[d] = (unsigned int)(maxint[d] - minint[d] + 1);
    int bitsize;
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff) {
        for (int d = 0; d < 3; d++) bitsizeint[d] = bits_of(sizeint[d]);
        bitsize = 0;
    } else {
        bitsize = bits_of_product(sizeint);
    }
    int smallidx = kFirstIdx;
    while (smallidx < kLastIdx - 1 && kMagic[smallidx] < mindiff) smallidx++;
    xdr_put_u32(o, (uint32_t)smallidx);
    XtcParams P;
    for (int d = 0; d < 3; d++) { P.minint[d] = minint[d]; P.sizeint[d] = sizeint[d]; P.bitsizeint[d] = bitsizeint[d]; }
    P.bitsize = bitsize;
    // the table's last entry bounds both indices (frames with > 16777216 between all neighbours)
    P.maxidx = (kLastIdx - 1 < smallidx + 8) ? kLastIdx - 1 : smallidx + 8;
    P.minidx = P.maxidx - 8;
    P.larger = kMagic[P.maxidx] / 2;
    XtcState st0;
    st0.i = 0; st0.smallidx = smallidx; st0.prevrun = -1; st0.prev[0] = st0.prev[1] = st0.prev[2] = 0;
    st0.smaller = kMagic[(kFirstIdx > smallidx - 1) ? kFirstIdx : smallidx - 1] / 2;
    st0.smallnum = kMagic[smallidx] / 2;
    static thread_local std::vector<unsigned char> bits_scratch;
    BitWriter w;
    w.buf.swap(bits_scratch);
    w.reserve_bytes((size_t)natoms * 16 + 64);     // > 3 x 32 bits + run flags per atom
    const int T = (nthreads > 1 && natoms >= 8192) ? (nthreads > 16 ? 16 : nthreads) : 1;
    if (T == 1) {
        xtc_encode_range(ip, natoms, P, st0, natoms, &w);
    } else {
        // Speculative segments: thread k > 0 runs the encoder's state machine, without output, over the kWarm atoms
        // ahead of its segment (the adaptive small-number index and the run boundaries settle within a few hundred
        // atoms whatever they started from), then encodes its segment. A segment is kept when the state its
        // predecessor really ended in is the state it assumed; otherwise it is encoded again from the true state, so
        // the bytes are those of the one-thread encoder in every case.
        const int kWarm = g_xtc_warm;
        long long respeculated = 0;
        std::vector<XtcState> assumed(T), ended(T);
        static thread_local std::vector<BitWriter> segw;
        if ((int)segw.size() < T) segw.resize(T);
        for (int k = 1; k < T; k++) { segw[k].len = 0; segw[k].acc = 0; segw[k].nacc = 0; segw[k].reserve_bytes(((size_t)natoms / T + 64) * 16 + 64); }
        BitWriter *const seg = segw.data();        // (a thread_local name inside the parallel region would mean each worker's own, empty, vector)
        BitWriter *const w0 = &w;
        const double t_par0 = now_ms();
        double worker_ms[16] = { 0 };
        run_slices(T, [&](int k) {
            const double t_w0 = now_ms();
            const int b0 = (int)((long long)natoms * k / T), b1 = (k + 1 == T) ? natoms : (int)((long long)natoms * (k + 1) / T);
            XtcState st = st0;
            if (k > 0) {
                st.i = b0 > kWarm ? b0 - kWarm : 0;
                if (st.i > 0) { st.prev[0] = ip[3 * (size_t)(st.i - 1)]; st.prev[1] = ip[3 * (size_t)(st.i - 1) + 1]; st.prev[2] = ip[3 * (size_t)(st.i - 1) + 2]; }
                xtc_encode_range(ip, natoms, P, st, b0, nullptr);
            }
            assumed[k] = st;
            xtc_encode_range(ip, natoms, P, st, b1, k == 0 ? w0 : &seg[k]);
            ended[k] = st;
            worker_ms[k] = now_ms() - t_w0;
        });
        const double t_par1 = now_ms();
        for (int k = 1; k < T; k++) {
            const int b1 = (k + 1 == T) ? natoms : (int)((long long)natoms * (k + 1) / T);
            if (ended[k - 1].i >= b1 && k + 1 < T) {       // the predecessor ran past this whole segment (one long run): nothing of it is left
                ended[k] = ended[k - 1];
                continue;
            }
            if (ended[k - 1] == assumed[k]) {
                w.append(seg[k]);
            } else {
                respeculated++;
                XtcState st = ended[k - 1];
                xtc_encode_range(ip, natoms, P, st, b1, &w);
                ended[k] = st;
            }
        }
        g_xtc_respeculated += respeculated;
        g_xtc_phase[0] = t_par1 - t_par0; g_xtc_phase[1] = now_ms() - t_par1;
        g_xtc_phase[3] = 0; for (int k = 0; k < T; k++) if (worker_ms[k] > g_xtc_phase[3]) g_xtc_phase[3] = worker_ms[k];
        for (int k = 0; k < 16; k++) g_xtc_phase[4 + k] = k < T ? worker_ms[k] : 0.0;
    }
    const size_t nbytes = w.finish();
    xdr_put_u32(o, (uint32_t)nbytes);
    o.insert(o.end(), w.buf.begin(), w.buf.begin() + (long)nbytes);
    bits_scratch.swap(w.buf);
    for (size_t k = nbytes; k % 4 != 0; k++) o.push_back(0);
    return TRJ_OK;
}

// Encode one frame into a caller-owned buffer; returns the byte count or a negative error.
// cap must be at least trj_xtc_frame_bound(natoms).
long long trj_xtc_frame_bound(int natoms) { return 128 + (long long)natoms * 16; }

// nthreads > 1 encodes segments of a large frame on that many threads (same bytes as one thread).
long long trj_xtc_encode_frame_mt(unsigned char *out, long long cap, int natoms, long long step, float time, const float *box9,
                                  const float *x, float precision, int nthreads)
{
    std::vector<unsigned char> o;
    o.reserve((size_t)natoms * 5 + 128);
    xdr_put_u32(o, 1995u);
    xdr_put_u32(o, (uint32_t)natoms);
    xdr_put_u32(o, (uint32_t)(int32_t)step);
    xdr_put_f32(o, time);
    for (int k = 0; k < 9; k++) xdr_put_f32(o, box9[k]);
    const int rc = xtc_encode_coords(o, natoms, x, precision, nthreads);
    if (rc != TRJ_OK) return rc;
    if ((long long)o.size() > cap) return TRJ_ERR_RANGE;
    memcpy(out, o.data(), o.size());
    return (long long)o.size();
}

long long trj_xtc_encode_frame(unsigned char *out, long long cap, int natoms, long long step, float time, const float *box9,
                               const float *x, float precision)
{
    return trj_xtc_encode_frame_mt(out, cap, natoms, step, time, box9, x, precision, 1);
}

// The same frame from coordinates that are rounded already (the engine's capture kernel multiplies by the precision
// and rounds on the GPU): ix holds 3 integers per atom, stats8 = extent minimum (3), extent maximum (3), smallest sum of
// absolute differences between neighbouring atoms, and a flag for a coordinate that is not finite or out of range.
// The bytes are those trj_xtc_encode_frame_mt gives for the float coordinates. Frames of fewer than 10 atoms are
// stored as floats by the format and cannot come this way.
long long trj_xtc_encode_frame_q(unsigned char *out, long long cap, int natoms, long long step, float time, const float *box9,
                                 const int *ix, const int *stats8, float precision, int nthreads)
{
    if (natoms <= 9) return TRJ_ERR_RANGE;
    if (stats8[7] != 0) return TRJ_ERR_RANGE;
    std::vector<unsigned char> o;
    o.reserve((size_t)natoms * 5 + 128);
    xdr_put_u32(o, 1995u);
    xdr_put_u32(o, (uint32_t)natoms);
    xdr_put_u32(o, (uint32_t)(int32_t)step);
    xdr_put_f32(o, time);
    for (int k = 0; k < 9; k++) xdr_put_f32(o, box9[k]);
    xdr_put_u32(o, (uint32_t)natoms);
    if (precision <= 0) precision = 1000.0f;
    xdr_put_f32(o, precision);
    const int rc = xtc_pack_coords(o, natoms, ix, stats8, stats8 + 3, stats8[6], nthreads);
    if (rc != TRJ_OK) return rc;
    if ((long long)o.size() > cap) return TRJ_ERR_RANGE;
    memcpy(out, o.data(), o.size());
    return (long long)o.size();
}

int trj_xtc_append(const char *path, int truncate, int natoms, long long step, float time, const float *box9, const float *x,
                   float precision)
{
    std::vector<unsigned char> buf((size_t)trj_xtc_frame_bound(natoms));
    const long long n = trj_xtc_encode_frame(buf.data(), (long long)buf.size(), natoms, step, time, box9, x, precision);
    if (n < 0) return (int)n;
    FILE *f = fopen(path, truncate ? "wb" : "ab");
    if (!f) return TRJ_ERR_IO;
    const size_t wr = fwrite(buf.data(), 1, (size_t)n, f);
    const int cl = fclose(f);
    return (wr == (size_t)n && cl == 0) ? TRJ_OK : TRJ_ERR_IO;
}

// Decode the frame that starts at `data`; *used receives its length. x must hold 3*natoms floats
// (query natoms with x == NULL first: the header is parsed and *natoms set).
int trj_xtc_decode_frame(const unsigned char *data, long long len, long long *used, int *natoms, long long *step, float *time,
                         float *box9, float *x, float *precision)
{
    if (len == 0) return TRJ_EOF;
    if (len < 56) return TRJ_ERR_FORMAT;
    if (be32(data) != 1995u) return TRJ_ERR_FORMAT;
    const int n = (int)be32(data + 4);
    if (natoms) *natoms = n;
    if (step) *step = (int32_t)be32(data + 8);
    if (time) *time = bef32(data + 12);
    if (box9) for (int k = 0; k < 9; k++) box9[k] = bef32(data + 16 + 4 * k);
    if ((int)be32(data + 52) != n) return TRJ_ERR_FORMAT;
    long long pos = 56;
    if (n <= 9) {
        if (len < pos + 12LL * n) return TRJ_ERR_FORMAT;
        if (x) for (int i = 0; i < 3 * n; i++) x[i] = bef32(data + pos + 4 * i);
        if (precision) *precision = 0.0f;
        if (used) *used = pos + 12LL * n;
        return TRJ_OK;
    }
    if (len < pos + 36) return TRJ_ERR_FORMAT;
    const float prec = bef32(data + pos); pos += 4;
    if (precision) *precision = prec;
    int minint[3], maxint[3];
    for (int d = 0; d < 3; d++) { minint[d] = (int32_t)be32(data + pos); pos += 4; }
    for (int d = 0; d < 3; d++) { maxint[d] = (int32_t)be32(data + pos); pos += 4; }
    int smallidx = (int)be32(data + pos); pos += 4;
    const long long nbytes = be32(data + pos); pos += 4;
    const long long padded = (nbytes + 3) / 4 * 4;
    if (len < pos + padded) return TRJ_ERR_FORMAT;
    if (used) *used = pos + padded;
    if (!x) return TRJ_OK;
    if (smallidx < kFirstIdx || smallidx >= kLastIdx) return TRJ_ERR_FORMAT;
    unsigned int sizeint[3], bitsizeint[3] = { 0, 0, 0 };
    for (int d = 0; d < 3; d++) sizeint[d] = (unsigned int)(maxint[d] - minint[d] + 1);
    int bitsize;
    if ((sizeint[0] | sizeint[1] | sizeint[2]) > 0xffffff) {
        for (int d = 0; d < 3; d++) bitsizeint[d] = bits_of(sizeint[d]);
        bitsize = 0;
    } else bitsize = bits_of_product(sizeint);
    int smaller = kMagic[(kFirstIdx > smallidx - 1) ? kFirstIdx : smallidx - 1] / 2;
    int smallnum = kMagic[smallidx] / 2;
    unsigned int sizesmall[3] = { (unsigned)kMagic[smallidx], (unsigned)kMagic[smallidx], (unsigned)kMagic[smallidx] };
    BitReader r(data + pos, (size_t)nbytes);
    const float inv = 1.0f / prec;
    int run = 0, i = 0;
    float *lfp = x;
    while (i < n) {
        int c[3], prev[3];
        if (bitsize == 0) { c[0] = (int)r.get(bitsizeint[0]); c[1] = (int)r.get(bitsizeint[1]); c[2] = (int)r.get(bitsizeint[2]); }
        else get_triple(r, bitsize, sizeint, c);
        i++;
        c[0] += minint[0]; c[1] += minint[1]; c[2] += minint[2];
        prev[0] = c[0]; prev[1] = c[1]; prev[2] = c[2];
        const unsigned flag = r.get(1);
        int is_smaller = 0;
        if (flag == 1) {
            run = (int)r.get(5);
            is_smaller = run % 3;
            run -= is_smaller;
            is_smaller--;
        }
        if (r.bad) return TRJ_ERR_FORMAT;
        if (run > 0) {
            if (i + run / 3 > n) return TRJ_ERR_FORMAT;
            for (int k = 0; k < run; k += 3) {
                int t[3];
                get_triple(r, smallidx, sizesmall, t);
                i++;
                t[0] += prev[0] - smallnum; t[1] += prev[1] - smallnum; t[2] += prev[2] - smallnum;
                if (k == 0) {
                    // undo the interchange: the run's first small atom was the frame's earlier atom
                    *lfp++ = t[0] * inv; *lfp++ = t[1] * inv; *lfp++ = t[2] * inv;
                    *lfp++ = c[0] * inv; *lfp++ = c[1] * inv; *lfp++ = c[2] * inv;
                } else {
                    *lfp++ = t[0] * inv; *lfp++ = t[1] * inv; *lfp++ = t[2] * inv;
                }
                prev[0] = t[0]; prev[1] = t[1]; prev[2] = t[2];
            }
        } else {
            *lfp++ = c[0] * inv; *lfp++ = c[1] * inv; *lfp++ = c[2] * inv;
        }
        smallidx += is_smaller;
        if (is_smaller < 0) { smallnum = smaller; smaller = smallidx > kFirstIdx ? kMagic[smallidx - 1] / 2 : 0; }
        else if (is_smaller > 0) { smaller = smallnum; smallnum = kMagic[smallidx] / 2; }
        sizesmall[0] = sizesmall[1] = sizesmall[2] = (unsigned)kMagic[smallidx];
        if (r.bad) return TRJ_ERR_FORMAT;
    }
    return TRJ_OK;
}

// ---------------------------------------------------------------- trr
// Single-precision frame: header (magic 1993, "GMX_trn_file", block sizes, natoms, step, t,
// lambda) then box, x, and optionally v and f.
long long trj_trr_frame_bound(int natoms) { return 256 + 36LL * natoms; }

long long trj_trr_encode_frame(unsigned char *out, long long cap, int natoms, long long step, float time, float lambda,
                               const float *box9, const float *x, const float *v, const float *f)
{
    std::vector<unsigned char> o;
    o.reserve(128 + (size_t)natoms * 12 * ((x ? 1 : 0) + (v ? 1 : 0) + (f ? 1 : 0)));
    static const char version[] = "GMX_trn_file";
    xdr_put_u32(o, 1993u);
    xdr_put_u32(o, 13u);
    xdr_put_u32(o, 12u);
    for (int k = 0; k < 12; k++) o.push_back((unsigned char)version[k]);
    const uint32_t vec = (uint32_t)natoms * 12u;
    // ir, e, box, vir, pres, top, sym, x, v, f
    const uint32_t sizes[10] = { 0, 0, box9 ? 36u : 0u, 0, 0, 0, 0, x ? vec : 0u, v ? vec : 0u, f ? vec : 0u };
    for (int k = 0; k < 10; k++) xdr_put_u32(o, sizes[k]);
    xdr_put_u32(o, (uint32_t)natoms);
    xdr_put_u32(o, (uint32_t)(int32_t)step);
    xdr_put_u32(o, 0u);                       // nre
    xdr_put_f32(o, time);
    xdr_put_f32(o, lambda);
    if (box9) for (int k = 0; k < 9; k++) xdr_put_f32(o, box9[k]);
    const float *blocks[3] = { x, v, f };
    for (const float *b : blocks) {
        if (!b) continue;
        const size_t at = o.size();
        o.resize(at + (size_t)natoms * 12);
        unsigned char *p = o.data() + at;
        for (size_t i = 0; i < 3 * (size_t)natoms; i++) {
            uint32_t u; memcpy(&u, &b[i], 4);
            p[4 * i] = (unsigned char)(u >> 24); p[4 * i + 1] = (unsigned char)(u >> 16); p[4 * i + 2] = (unsigned char)(u >> 8); p[4 * i + 3] = (unsigned char)u;
        }
    }
    if ((long long)o.size() > cap) return TRJ_ERR_RANGE;
    memcpy(out, o.data(), o.size());
    return (long long)o.size();
}

int trj_trr_append(const char *path, int truncate, int natoms, long long step, float time, float lambda, const float *box9,
                   const float *x, const float *v, const float *f)
{
    std::vector<unsigned char> buf((size_t)trj_trr_frame_bound(natoms));
    const long long n = trj_trr_encode_frame(buf.data(), (long long)buf.size(), natoms, step, time, lambda, box9, x, v, f);
    if (n < 0) return (int)n;
    FILE *fp = fopen(path, truncate ? "wb" : "ab");
    if (!fp) return TRJ_ERR_IO;
    const size_t wr = fwrite(buf.data(), 1, (size_t)n, fp);
    const int cl = fclose(fp);
    return (wr == (size_t)n && cl == 0) ? TRJ_OK : TRJ_ERR_IO;
}

// Decode a single-precision trr frame. has[3] reports which of x, v, f were present.
int trj_trr_decode_frame(const unsigned char *data, long long len, long long *used, int *natoms, long long *step, float *time,
                         float *lambda, float *box9, float *x, float *v, float *f, int *has)
{
    if (len == 0) return TRJ_EOF;
    if (len < 24 + 13 * 4 + 8) return TRJ_ERR_FORMAT;
    if (be32(data) != 1993u || be32(data + 4) != 13u || be32(data + 8) != 12u || memcmp(data + 12, "GMX_trn_file", 12) != 0) return TRJ_ERR_FORMAT;
    long long pos = 24;
    uint32_t sizes[10];
    for (int k = 0; k < 10; k++) { sizes[k] = be32(data + pos); pos += 4; }
    const int n = (int)be32(data + pos); pos += 4;
    if (natoms) *natoms = n;
    if (step) *step = (int32_t)be32(data + pos);
    pos += 4;
    pos += 4;   // nre
    if (sizes[2] != 0 && sizes[2] != 36u) return TRJ_ERR_FORMAT;       // double precision files are not handled
    if (time) *time = bef32(data + pos);
    pos += 4;
    if (lambda) *lambda = bef32(data + pos);
    pos += 4;
    const long long body = (long long)sizes[0] + sizes[1] + sizes[2] + sizes[3] + sizes[4] + sizes[5] + sizes[6] + sizes[7] + sizes[8] + sizes[9];
    if (len < pos + body) return TRJ_ERR_FORMAT;
    if (used) *used = pos + body;
    pos += sizes[0] + sizes[1];
    if (sizes[2]) { if (box9) for (int k = 0; k < 9; k++) box9[k] = bef32(data + pos + 4 * k); pos += sizes[2]; }
    pos += sizes[3] + sizes[4] + sizes[5] + sizes[6];
    float *dst[3] = { x, v, f };
    for (int b = 0; b < 3; b++) {
        if (has) has[b] = sizes[7 + b] ? 1 : 0;
        if (sizes[7 + b]) {
            if (sizes[7 + b] != (uint32_t)n * 12u) return TRJ_ERR_FORMAT;
            if (dst[b]) for (long long i = 0; i < 3LL * n; i++) dst[b][i] = bef32(data + pos + 4 * i);
            pos += sizes[7 + b];
        }
    }
    return TRJ_OK;
}

long long trj_xtc_respeculated(void) { return g_xtc_respeculated.load(); }
// Diagnostic: where the calling thread's last multi-threaded frame spent its time (ms): the parallel segments, the merge
// of the segments, (unused), the slowest segment's own time.
void trj_xtc_last_phases(double *out20) { for (int k = 0; k < 20; k++) out20[k] = g_xtc_phase[k]; }
// Test hook: length of the warm-up walk (0 makes nearly every segment start from a wrong guess, which exercises the
// path that encodes a segment again from the true state). Returns the previous value.
int trj_xtc_set_warmup(int atoms) { const int old = g_xtc_warm; g_xtc_warm = atoms < 0 ? 0 : atoms; return old; }

int trj_version(void) { return 2; }

}  // extern "C"
