// gfx950 kernels of step 4's front end: external contours of the final foreground mask
// and the per-blob quantities GOFPID._find_blob consumes, computed where the mask is.
//
// Replaces  helpers.find_contours(mask)  = cv.findContours(mask, RETR_EXTERNAL,
// CHAIN_APPROX_NONE) [REF helpers.py:125-150] followed by cv.contourArea(contour) >= 100
// [REF pipeline.py:555-572] and the cv.moments / np.max calls of get_bottom / get_center
// [REF helpers.py:153-168], so that only blob records and the contour points of blobs
// that pass the area filter cross PCIe (KBs) instead of the whole mask (2 MB at 1080p).
//
// Border following is sequential; this is its parallel restatement.  A CRACK is a
// (foreground pixel, side) pair whose 4-neighbour across that side is background (outside
// the image = background).  Walking the cracks with the background on the right-hand side
// and preferring the diagonal neighbour at every vertex (8-connected foreground) gives a
// successor function that is a permutation of the cracks; its cycles are exactly the
// borders Suzuki-Abe border following traces, and the pixels of successive cracks, with
// runs of one pixel merged, are exactly the points it emits, in its order, starting from
// the run that holds the north crack of the border's raster-first pixel
// (checked against the host follower gof_contours.cpp on random masks: tests/).  Hence
//   * cycle ids        = union-find over crack -> successor (root = smallest crack key
//                        = north crack of the raster-first pixel),
//   * contour polygon  = sum over cracks of cross(pixel, pixel of successor): the shoelace
//                        sums a00 / a10 / a01 of cv.moments and cv.contourArea are sums of
//                        LOCAL terms (exact in int64), bounding box likewise,
//   * outer vs hole    = sign of a00 (holes run the other way round),
//   * RETR_EXTERNAL    = nesting depth 0: outer-border W cracks minus outer-border E
//                        cracks to the left of the start pixel on its row,
//   * point order      = list ranking (pointer jumping) along the cycle, only for blobs
//                        that pass the area filter.
// All integer work; no floating point on the device.
#include "gof_kernels.h"

#include <limits.h>

namespace gofk {

namespace {

// side codes: the north crack of a pixel has the smallest key of its pixel
constexpr unsigned SN = 0, SW = 1, SS = 2, SE = 3;
constexpr uint32_t NIL = 0xffffffffu;

// Parent pointers are read and rewritten by workgroups on different XCDs at the same
// time.  The per-XCD L2s are not coherent with each other and a compare-and-swap executes
// at the memory side, so a plain (even sc1) load may keep returning an L2 copy that a
// CAS elsewhere has long replaced -- a find would then present a stale root to its own CAS
// for ever.  Every access to a parent pointer is therefore itself a memory-side atomic:
// the read is a fetch-or of zero, the path-halving write a compare-and-swap.
__device__ __forceinline__ uint32_t ld_parent(uint32_t *p)
{
    return __hip_atomic_fetch_or(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cas_parent(uint32_t *p, uint32_t expect, uint32_t v)
{
    return __hip_atomic_compare_exchange_strong(p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
}

// union-find over cracks, root = smallest key.  Keys strictly decrease along parent
// pointers (a root is only ever linked under a root of smaller key; path halving
// re-points to an ancestor), so concurrent finds / unions cannot form a loop.  Every
// loop is bounded all the same: a bound that is hit raises kFlagInternal instead of
// spinning (the host then falls back to the mask, and the tests fail).
constexpr int kMaxFindSteps = 1 << 22, kMaxUnionTries = 1 << 16;

__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t i, uint32_t *flags)
{
    for (int it = 0; it < kMaxFindSteps; it++) {
        const uint32_t p = ld_parent(parent + i);
        if (p == i)
            return i;
        const uint32_t g = ld_parent(parent + p);
        if (g == p)
            return p;
        (void)cas_parent(parent + i, p, g);   // path halving (lost races change nothing)
        i = g;
    }
    atomicOr(flags, (uint32_t)kFlagInternal);
    return i;
}

__device__ __forceinline__ void uf_union(uint32_t *parent, const uint32_t *key, uint32_t a, uint32_t b,
                                         uint32_t *flags)
{
    for (int it = 0; it < kMaxUnionTries; it++) {
        a = uf_find(parent, a, flags);
        b = uf_find(parent, b, flags);
        if (a == b)
            return;
        if (key[a] > key[b]) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        // a has the smaller key: hang b under it, if b is still a root
        if (cas_parent(parent + b, b, a))
            return;
    }
    atomicOr(flags, (uint32_t)kFlagInternal);
}

struct Nbr {   // 3 x 3 neighbourhood of a foreground pixel as booleans (outside = background)
    bool n, s, w, e, nw, ne, sw, se;
};

__device__ __forceinline__ Nbr load_nbr(const uint8_t *m, int x, int y, int W, int H)
{
    Nbr b;
    const bool up = y > 0, dn = y < H - 1, lf = x > 0, rt = x < W - 1;
    const uint8_t *r = m + (size_t)y * W + x;
    b.n = up && r[-W] != 0;
    b.s = dn && r[W] != 0;
    b.w = lf && r[-1] != 0;
    b.e = rt && r[1] != 0;
    b.nw = up && lf && r[-W - 1] != 0;
    b.ne = up && rt && r[-W + 1] != 0;
    b.sw = dn && lf && r[W - 1] != 0;
    b.se = dn && rt && r[W + 1] != 0;
    return b;
}

__device__ __forceinline__ unsigned sides_of(const Nbr &b)
{
    return (b.n ? 0u : 1u << SN) | (b.w ? 0u : 1u << SW) | (b.s ? 0u : 1u << SS) | (b.e ? 0u : 1u << SE);
}

// ---- K1: find the cracks, give each a slot -------------------------------------------------
// Tile 64 x 4, a wave per row, a pixel per lane; a wave whose 64 pixels are all zero leaves
// at once (most of a surveillance mask).  Slots are handed out per wave with one atomic.
__global__ __launch_bounds__(256) void blob_scan_kernel(const uint8_t *__restrict__ mask, BlobBuffers B)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + wave, img = blockIdx.z;
    const int W = B.w, H = B.h;
    const size_t N = (size_t)W * H;
    const uint8_t *m = mask + (size_t)img * N;
    const bool in = x < W && y < H;
    const bool fg = in && m[(size_t)y * W + x] != 0;
    if (__builtin_amdgcn_ballot_w64(fg) == 0)
        return;
    unsigned sides = 0;
    if (fg)
        sides = sides_of(load_nbr(m, x, y, W, H));
    const unsigned cnt = __builtin_popcount(sides);
    const uint64_t b0 = __builtin_amdgcn_ballot_w64(cnt & 1u), b1 = __builtin_amdgcn_ballot_w64(cnt & 2u),
                   b2 = __builtin_amdgcn_ballot_w64(cnt & 4u);
    const unsigned total = __builtin_popcountll(b0) + 2 * __builtin_popcountll(b1) + 4 * __builtin_popcountll(b2);
    if (total == 0)
        return;
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    const unsigned pre = __builtin_popcountll(b0 & lt) + 2 * __builtin_popcountll(b1 & lt) +
                         4 * __builtin_popcountll(b2 & lt);
    uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    uint32_t base = 0;
    if (lane == 0)
        base = atomicAdd(cnts + kCntCracks, total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (cnt == 0)
        return;
    const uint32_t ci = base + pre;
    if (ci + cnt > B.cap) {
        cnts[kCntFlags] = kFlagCrackOverflow;   // (same value from every writer)
        return;
    }
    const uint32_t pix = (uint32_t)y * W + x;
    B.cbase[(size_t)img * N + pix] = (ci << 4) | sides;
    uint32_t *key = B.ckey + (size_t)img * B.cap, *parent = B.cparent + (size_t)img * B.cap;
    unsigned r = 0;
#pragma unroll
    for (unsigned s = 0; s < 4; s++) {
        if (sides & (1u << s)) {
            key[ci + r] = pix * 4u + s;
            parent[ci + r] = ci + r;
            r++;
        }
    }
}

// slot of crack (pixel entry e, side s)
__device__ __forceinline__ uint32_t slot_of(uint32_t e, unsigned s)
{
    return (e >> 4) + __builtin_popcount((e & 15u) & ((1u << s) - 1u));
}

// ---- K2: successor of every crack, and union with it -----------------------------------------
// Walking direction per side (background on the right): W south, S east, E north, N west.
// At the far vertex: the diagonal pixel ahead-right, if foreground, takes over (turn right,
// onto ITS previous side); else the pixel ahead (same side); else turn left (next side,
// same pixel).
__global__ __launch_bounds__(256) void blob_link_kernel(const uint8_t *__restrict__ mask, BlobBuffers B)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + wave, img = blockIdx.z;
    const int W = B.w, H = B.h;
    const size_t N = (size_t)W * H;
    const uint8_t *m = mask + (size_t)img * N;
    uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    if (cnts[kCntFlags] != 0)
        return;
    if (!(x < W && y < H) || m[(size_t)y * W + x] == 0)
        return;
    const Nbr b = load_nbr(m, x, y, W, H);
    const unsigned sides = sides_of(b);
    if (sides == 0)
        return;
    const uint32_t *cb = B.cbase + (size_t)img * N;
    const uint32_t *key = B.ckey + (size_t)img * B.cap;
    uint32_t *succ = B.csucc + (size_t)img * B.cap, *parent = B.cparent + (size_t)img * B.cap;
    const uint32_t pix = (uint32_t)y * W + x;
    const uint32_t e = cb[pix];
    // per side: (ahead-right pixel is fg, its offset, its side) / (ahead pixel is fg, offset) / next side
    //   W: walk S, right W -> ahead-right SW (side N), ahead S
    //   S: walk E, right S -> ahead-right SE (side W), ahead E
    //   E: walk N, right E -> ahead-right NE (side S), ahead N
    //   N: walk W, right N -> ahead-right NW (side E), ahead W
    auto link = [&](unsigned s, bool diag, int doff, unsigned dside, bool ahead, int aoff, unsigned nside) {
        if (!(sides & (1u << s)))
            return;
        const uint32_t ci = slot_of(e, s);
        uint32_t cn;
        if (diag)
            cn = slot_of(cb[(int)pix + doff], dside);
        else if (ahead)
            cn = slot_of(cb[(int)pix + aoff], s);
        else
            cn = slot_of(e, nside);
        succ[ci] = cn;
        uf_union(parent, key, ci, cn, cnts + kCntFlags);
    };
    link(SW, b.sw, W - 1, SN, b.s, W, SS);
    link(SS, b.se, W + 1, SW, b.e, 1, SE);
    link(SE, b.ne, -W + 1, SS, b.n, -W, SN);
    link(SN, b.nw, -W - 1, SE, b.w, -1, SW);
}

// ---- K3: flatten; every root takes an accumulator slot ----------------------------------------
__global__ __launch_bounds__(256) void blob_roots_kernel(BlobBuffers B)
{
    const int img = blockIdx.y;
    uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    if (cnts[kCntFlags] != 0)
        return;
    const uint32_t nc = cnts[kCntCracks];
    uint32_t *parent = B.cparent + (size_t)img * B.cap, *aux = B.caux + (size_t)img * B.cap;
    BlobAcc *acc = B.acc + (size_t)img * (B.cap / 4);
    for (uint32_t ci = blockIdx.x * 256 + threadIdx.x; ci < nc; ci += gridDim.x * 256) {
        const uint32_t r = uf_find(parent, ci, cnts + kCntFlags);
        if (r != ci)   // memory-side like every other access to it; the last word on this pointer
            (void)__hip_atomic_exchange(parent + ci, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (r == ci) {
            const uint32_t slot = atomicAdd(cnts + kCntRoots, 1u);   // <= cracks / 4 <= cap / 4
            aux[ci] = slot;
            BlobAcc a;
            a.a00 = a.a10 = a.a01 = 0;
            a.xmin = a.ymin = INT_MAX;
            a.xmax = a.ymax = INT_MIN;
            a.trans = 0;
            a.root = ci;
            a.out = -1;
            acc[slot] = a;
        }
    }
}

__device__ __forceinline__ long long wave_sum64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_min32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int u = __shfl_xor(v, o, 64);
        v = u < v ? u : v;
    }
    return v;
}
__device__ __forceinline__ int wave_max32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int u = __shfl_xor(v, o, 64);
        v = u > v ? u : v;
    }
    return v;
}

// ---- K4: shoelace sums, transitions and bounding box per cycle ---------------------------------
// Slots were handed out wave by wave in K1, so the 64 cracks a wave takes here are
// neighbours in the image and nearly always of one cycle: the wave reduces per distinct
// root and issues one set of atomics for each.
__global__ __launch_bounds__(256) void blob_stats_kernel(BlobBuffers B)
{
    const int img = blockIdx.y;
    const uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    if (cnts[kCntFlags] != 0)
        return;
    const uint32_t nc = cnts[kCntCracks];
    const uint32_t *key = B.ckey + (size_t)img * B.cap, *succ = B.csucc + (size_t)img * B.cap,
                   *parent = B.cparent + (size_t)img * B.cap, *aux = B.caux + (size_t)img * B.cap;
    BlobAcc *acc = B.acc + (size_t)img * (B.cap / 4);
    const uint32_t W = (uint32_t)B.w;
    const int lane = threadIdx.x & 63;
    const uint32_t start = (blockIdx.x * 256 + threadIdx.x) & ~63u;   // wave-uniform loop bounds
    for (uint32_t c0 = start; c0 < nc; c0 += gridDim.x * 256) {
        const uint32_t ci = c0 + lane;
        const bool valid = ci < nc;
        uint32_t slot = NIL;
        long long d00 = 0, d10 = 0, d01 = 0;
        int tr = 0, px = 0, py = 0;
        if (valid) {
            slot = aux[parent[ci]];
            const uint32_t p0 = key[ci] >> 2, p1 = key[succ[ci]] >> 2;
            px = (int)(p0 % W);
            py = (int)(p0 / W);
            if (p0 != p1) {
                const int x1 = (int)(p1 % W), y1 = (int)(p1 / W);
                const long long cr = (long long)px * y1 - (long long)x1 * py;
                d00 = cr;
                d10 = cr * (px + x1);
                d01 = cr * (py + y1);
                tr = 1;
            }
        }
        uint64_t todo = __builtin_amdgcn_ballot_w64(valid);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t s0 = (uint32_t)__shfl((int)slot, leader, 64);
            const bool mine = valid && slot == s0;
            const long long s00 = wave_sum64(mine ? d00 : 0), s10 = wave_sum64(mine ? d10 : 0),
                            s01 = wave_sum64(mine ? d01 : 0);
            const int st = wave_sum32(mine ? tr : 0);
            const int x0 = wave_min32(mine ? px : INT_MAX), x1 = wave_max32(mine ? px : INT_MIN),
                      y0 = wave_min32(mine ? py : INT_MAX), y1 = wave_max32(mine ? py : INT_MIN);
            if (lane == leader) {
                BlobAcc *a = acc + s0;
                if (s00)
                    atomicAdd((unsigned long long *)&a->a00, (unsigned long long)s00);
                if (s10)
                    atomicAdd((unsigned long long *)&a->a10, (unsigned long long)s10);
                if (s01)
                    atomicAdd((unsigned long long *)&a->a01, (unsigned long long)s01);
                if (st)
                    atomicAdd(&a->trans, st);
                atomicMin(&a->xmin, x0);
                atomicMax(&a->xmax, x1);
                atomicMin(&a->ymin, y0);
                atomicMax(&a->ymax, y1);
            }
            todo &= ~__builtin_amdgcn_ballot_w64(mine);
        }
    }
}

// ---- K5: select, order, rank, emit -- one 1024-thread workgroup per image -----------------------
constexpr int FT = 1024;           // threads
constexpr int kMaxBlobs = 4096;    // returned borders per image (more: kFlagBlobOverflow, the host takes the mask)

__global__ __launch_bounds__(FT) void blob_finish_kernel(const uint8_t *__restrict__ mask, BlobBuffers B,
                                                         double min_area)
{
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    uint32_t *out = B.out + (size_t)img * B.out_words;
    BlobHeader *hdr = (BlobHeader *)out;
    if (cnts[kCntFlags] != 0) {
        if (tid == 0) {
            hdr->n_blobs = 0;
            hdr->n_points = 0;
            hdr->flags = cnts[kCntFlags];
            hdr->n_cracks = cnts[kCntCracks];
        }
        return;
    }
    const int W = B.w, H = B.h;
    const size_t N = (size_t)W * H;
    const uint8_t *m = mask + (size_t)img * N;
    const uint32_t nc = cnts[kCntCracks], nroots = cnts[kCntRoots];
    const uint32_t *cb = B.cbase + (size_t)img * N;
    const uint32_t *key = B.ckey + (size_t)img * B.cap, *succ = B.csucc + (size_t)img * B.cap,
                   *parent = B.cparent + (size_t)img * B.cap, *aux = B.caux + (size_t)img * B.cap;
    uint32_t *cpos = B.cpos + (size_t)img * B.cap, *rsel = B.rsel + (size_t)img * B.cap,
             *rnxt = B.rnxt + (size_t)img * B.cap, *rsum = B.rsum + (size_t)img * B.cap;
    BlobAcc *acc = B.acc + (size_t)img * (B.cap / 4);

    __shared__ uint32_t acc_slot[kMaxBlobs];   // accepted borders: accumulator slot, by arrival
    __shared__ uint32_t acc_key[kMaxBlobs];
    __shared__ uint32_t by_rank[kMaxBlobs];    // accumulator slot by rank
    __shared__ uint32_t n_cand, n_acc, n_sel, flags;
    uint32_t *cand = rsel;                     // (the ranking scratch is free until phase D)
    if (tid == 0) {
        n_cand = n_acc = n_sel = 0;
        flags = 0;
    }
    __syncthreads();

    // A: outer borders (a00 <= 0: holes run the other way round) with polygon area >= min_area
    //    [REF pipeline.py:562-565: cv.contourArea(contour) >= area_min]
    for (uint32_t s = tid; s < nroots; s += FT) {
        const long long a00 = acc[s].a00;
        if (a00 <= 0 && (double)(-a00) * 0.5 >= min_area)
            cand[atomicAdd(&n_cand, 1u)] = s;   // roots <= cap / 4
    }
    __syncthreads();
    const uint32_t ncand = n_cand;

    // B: RETR_EXTERNAL = not inside another outer border.  Depth at the start pixel = outer
    //    borders entered minus outer borders left on the way there along its row.
    for (uint32_t j = wave; j < ncand; j += FT / 64) {
        const uint32_t slot = cand[j];
        const uint32_t rk = key[acc[slot].root];
        const int x0 = (int)((rk >> 2) % (uint32_t)W), y0 = (int)((rk >> 2) / (uint32_t)W);
        const uint8_t *row = m + (size_t)y0 * W;
        int d = 0;
        for (int x = lane; x < x0; x += 64) {
            if (row[x] == 0)
                continue;
            const bool wc = x == 0 || row[x - 1] == 0, ec = row[x + 1] == 0;   // x + 1 <= x0 < W
            if (!wc && !ec)
                continue;
            const uint32_t e = cb[(size_t)y0 * W + x];
            if (wc && acc[aux[parent[slot_of(e, SW)]]].a00 <= 0)
                d++;
            if (ec && acc[aux[parent[slot_of(e, SE)]]].a00 <= 0)
                d--;
        }
        d = wave_sum32(d);
        if (lane == 0 && d == 0) {
            const uint32_t k = atomicAdd(&n_acc, 1u);
            if (k < (uint32_t)kMaxBlobs) {
                acc_slot[k] = slot;
                acc_key[k] = rk;
            } else {
                flags = kFlagBlobOverflow;
            }
        }
    }
    __syncthreads();
    const uint32_t nacc = n_acc < (uint32_t)kMaxBlobs ? n_acc : (uint32_t)kMaxBlobs;

    // C: cv2 returns the contours in discovery (raster) order of their start pixels = key order
    for (uint32_t j = tid; j < nacc; j += FT) {
        const uint32_t k = acc_key[j];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < nacc; i++)
            rank += acc_key[i] < k ? 1u : 0u;
        by_rank[rank] = acc_slot[j];
    }
    __syncthreads();
    BlobRecord *recs = (BlobRecord *)(out + kBlobHeaderWords);
    if (tid == 0) {
        uint32_t off = 0;
        const uint32_t words = kBlobHeaderWords + nacc * kBlobRecordWords;
        const bool fits = words <= B.out_words && flags == 0;
        for (uint32_t r = 0; r < nacc; r++) {
            if (fits)
                recs[r].point_offset = (int)off;
            const int t = acc[by_rank[r]].trans;
            off += (uint32_t)(t > 0 ? t : 1);
        }
        if (!fits || off > B.out_words - words)
            flags |= kFlagPointOverflow;
        hdr->n_blobs = nacc;
        hdr->n_points = off;
        hdr->flags = flags;
        hdr->n_cracks = nc;
    }
    __syncthreads();
    if (flags != 0)
        return;   // (the host falls back to the mask itself)
    uint32_t *points = out + kBlobHeaderWords + nacc * kBlobRecordWords;
    for (uint32_t r = tid; r < nacc; r += FT) {
        BlobAcc *a = acc + by_rank[r];
        a->out = (int)r;
        const uint32_t rk = key[a->root];
        BlobRecord *q = recs + r;   // (point_offset is in place)
        q->x0 = (int)((rk >> 2) % (uint32_t)W);
        q->y0 = (int)((rk >> 2) / (uint32_t)W);
        q->xmin = a->xmin;
        q->ymin = a->ymin;
        q->xmax = a->xmax;
        q->ymax = a->ymax;
        q->n_points = a->trans > 0 ? a->trans : 1;
        q->a00 = a->a00;
        q->a10 = a->a10;
        q->a01 = a->a01;
    }
    __syncthreads();
    if (nacc == 0)
        return;

    // D: the cracks of accepted blobs, compacted
    for (uint32_t c0 = 0; c0 < nc; c0 += FT) {
        const uint32_t ci = c0 + tid;
        if (ci < nc && acc[aux[parent[ci]]].out >= 0) {
            const uint32_t pos = atomicAdd(&n_sel, 1u);
            rsel[pos] = ci;
            cpos[ci] = pos;
        }
    }
    __syncthreads();
    const uint32_t nsel = n_sel;

    // E: list ranking along each cycle, cut in front of its root crack.  rsum[j] ends as the
    //    number of pixel changes from crack j to the end of its cycle, so that crack j
    //    belongs to point (transitions of the cycle - rsum[j]).
    for (uint32_t j = tid; j < nsel; j += FT) {
        const uint32_t ci = rsel[j], cn = succ[ci];
        rsum[j] = (key[ci] >> 2) != (key[cn] >> 2) ? 1u : 0u;
        rnxt[j] = parent[cn] == cn ? NIL : cpos[cn];
    }
    __syncthreads();
    // pointer jumping, double-buffered: every step reads (sa, na) and writes (sb, nb)
    uint32_t *sa = rsum, *na = rnxt, *sb = B.rsum2 + (size_t)img * B.cap, *nb = B.rnxt2 + (size_t)img * B.cap;
    int go = 1;
    for (int step = 0; step < 34 && go; step++) {   // 2^32 list elements at most
        int any = 0;
        for (uint32_t j = tid; j < nsel; j += FT) {
            const uint32_t n = na[j];
            if (n != NIL) {
                sb[j] = sa[j] + sa[n];
                nb[j] = na[n];
                any = 1;
            } else {
                sb[j] = sa[j];
                nb[j] = NIL;
            }
        }
        go = __syncthreads_or(any);
        uint32_t *t = sa;
        sa = sb;
        sb = t;
        t = na;
        na = nb;
        nb = t;
    }
    if (go) {   // a list without an end: cannot happen (every cycle holds its root crack)
        if (tid == 0)
            hdr->flags |= kFlagInternal;
        return;
    }

    // F: every crack writes its pixel into its point's place (the cracks of one run write
    //    the same value); the run that closes the cycle on the start pixel is point 0 again
    for (uint32_t j = tid; j < nsel; j += FT) {
        const uint32_t ci = rsel[j];
        const BlobAcc *a = acc + aux[parent[ci]];
        const int tot = a->trans;
        const int idx = tot - (int)sa[j];
        const int np = tot > 0 ? tot : 1;
        if (idx < np) {
            const uint32_t p = key[ci] >> 2;
            points[(uint32_t)recs[a->out].point_offset + (uint32_t)idx] = (p % (uint32_t)W) | ((p / (uint32_t)W) << 16);
        }
    }
}

}  // namespace

hipError_t blob_extract(const uint8_t *mask, const BlobBuffers &B, double min_area, hipStream_t s)
{
    if (B.n_img < 1 || B.h < 1 || B.w < 1 || B.h > 65535 || B.w > 65535 || B.n_img > 65535 ||
        (B.h + 3) / 4 > 65535 || B.cap < 16 || B.cap > (1u << 28) || (B.cap & 3u) ||
        B.out_words < kBlobHeaderWords || (size_t)B.h * B.w > (1u << 30))
        return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(B.counters, 0, (size_t)B.n_img * kBlobCounters * sizeof(uint32_t), s);
    if (e != hipSuccess)
        return e;
    const dim3 g((unsigned)((B.w + 63) / 64), (unsigned)((B.h + 3) / 4), (unsigned)B.n_img);
    hipLaunchKernelGGL(blob_scan_kernel, g, dim3(256), 0, s, mask, B);
    hipLaunchKernelGGL(blob_link_kernel, g, dim3(256), 0, s, mask, B);
    // the crack count lives on the device: a fixed modest grid, grid-stride loops
    const dim3 g2(64, (unsigned)B.n_img);
    hipLaunchKernelGGL(blob_roots_kernel, g2, dim3(256), 0, s, B);
    hipLaunchKernelGGL(blob_stats_kernel, g2, dim3(256), 0, s, B);
    hipLaunchKernelGGL(blob_finish_kernel, dim3((unsigned)B.n_img), dim3(FT), 0, s, mask, B, min_area);
    return hipGetLastError();
}

}  // namespace gofk
