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

// Union for K2, where the trees are shallow (a crack points at its tile's local root,
// roots at roots of neighbouring tiles): both walks advance together, one round trip to the
// memory side per step instead of two, and nothing is re-pointed on the way (K3 flattens).
__device__ __forceinline__ void uf_union_pair(uint32_t *parent, const uint32_t *key, uint32_t a, uint32_t b,
                                              uint32_t *flags)
{
    for (int it = 0; it < kMaxUnionTries; it++) {
        for (int st = 0; st < kMaxFindSteps; st++) {
            const uint32_t pa = ld_parent(parent + a), pb = ld_parent(parent + b);
            if (pa == a && pb == b)
                break;
            a = pa;
            b = pb;
        }
        if (a == b)
            return;
        if (key[a] > key[b]) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
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

// ---- K0: which tiles hold any foreground ----------------------------------------------------
// A surveillance mask is mostly zeros.  One pass at streaming rate (a dword per lane, a
// wave per tile of 256 px x 16 rows, its 16 loads in flight together) lists the tiles that
// are not; the per-pixel kernels below run on the list only.  Entry: ty << 16 | tx.
constexpr int TX = 256, TY = 16;

__global__ __launch_bounds__(256) void blob_flag_kernel(const uint8_t *__restrict__ mask, BlobBuffers B, int aligned)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ty = blockIdx.y * 4 + wave, img = blockIdx.z;
    const int W = B.w, H = B.h;
    const int y0 = ty * TY;
    if (y0 >= H)
        return;
    const uint8_t *base = mask + (size_t)img * W * H;
    const int x = blockIdx.x * TX + 4 * lane;
    uint32_t any = 0;
#pragma unroll
    for (int r = 0; r < TY; r++) {
        const int y = y0 + r < H ? y0 + r : H - 1;   // (rows past the end re-read the last one)
        const uint8_t *row = base + (size_t)y * W;
        if (aligned) {             // W % 4 == 0, 4-byte aligned base: whole dwords inside the row
            if (x < W)
                any |= *(const uint32_t *)(row + x);
        } else {
#pragma unroll
            for (int b = 0; b < 4; b++)
                if (x + b < W)
                    any |= row[x + b];
        }
    }
    if (__builtin_amdgcn_ballot_w64(any != 0) == 0)
        return;
    if (lane == 0) {
        uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
        const uint32_t at = atomicAdd(cnts + kCntSegs, 1u);
        B.segs[(size_t)img * B.seg_cap + at] = ((uint32_t)ty << 16) | blockIdx.x;   // <= seg_cap by construction
    }
}

// slot of crack (pixel entry e, side s)
__device__ __forceinline__ uint32_t slot_of(uint32_t e, unsigned s)
{
    return (e >> 4) + __builtin_popcount((e & 15u) & ((1u << s) - 1u));
}

// Successor rule.  Walking direction per side (background on the right): W south, S east,
// E north, N west.  At the far vertex: the diagonal pixel ahead-right, if foreground, takes
// over (turn right, onto ITS previous side); else the pixel ahead (same side); else turn
// left (next side, same pixel).  Per side s of a pixel with neighbourhood b:
//   dx, dy = offset of the successor's pixel, ns = the successor's side
__device__ __forceinline__ void successor(const Nbr &b, unsigned s, int &dx, int &dy, unsigned &ns)
{
    bool diag, ahead;
    int ddx, ddy, adx, ady;
    unsigned dside, nside;
    switch (s) {
    case SW: diag = b.sw; ddx = -1; ddy = 1; dside = SN; ahead = b.s; adx = 0; ady = 1; nside = SS; break;
    case SS: diag = b.se; ddx = 1; ddy = 1; dside = SW; ahead = b.e; adx = 1; ady = 0; nside = SE; break;
    case SE: diag = b.ne; ddx = 1; ddy = -1; dside = SS; ahead = b.n; adx = 0; ady = -1; nside = SN; break;
    default: diag = b.nw; ddx = -1; ddy = -1; dside = SE; ahead = b.w; adx = -1; ady = 0; nside = SW; break;
    }
    if (diag) {
        dx = ddx;
        dy = ddy;
        ns = dside;
    } else if (ahead) {
        dx = adx;
        dy = ady;
        ns = s;
    } else {
        dx = dy = 0;
        ns = nside;
    }
}

// ---- K1: per tile -- find the cracks, give each a slot, link and union inside the tile --------
// One workgroup per listed tile.  The tile and a one-pixel frame sit in LDS; a thread owns
// 16 consecutive pixels of a row, so slots follow raster order inside the tile and the
// smallest local slot of a cycle part is its smallest key.  Links that stay inside the tile
// are united in LDS (no global atomics); links that leave it are left to K2.
constexpr int TBW = TX + 8;            // LDS row: 4 px to the left (one dword) .. 4 px to the right
// Border edges of a tile united in LDS.  A tile a morphological open / close has smoothed
// carries a few hundred; one with more than this (raw noise: up to two per pixel) leaves
// ALL its links to K2 (kBigTile in its list entry) -- so that the LDS footprint, and with it
// the number of tiles in flight per CU, is set by the common case, not by the worst.
constexpr int kTileCracks = 2048;
constexpr uint32_t kBigTile = 0x8000u;   // list entry: ty << 16 | kBigTile? | tx  (tx <= 255)

__device__ __forceinline__ uint32_t lds_find(uint32_t *lp, uint32_t i)
{
    for (;;) {
        const uint32_t p = lp[i];
        if (p == i)
            return i;
        const uint32_t g = lp[p];
        if (g == p)
            return p;
        lp[i] = g;   // path halving; keys (= local slots) strictly decrease along parents
        i = g;
    }
}

__global__ __launch_bounds__(256) void blob_tile_kernel(const uint8_t *__restrict__ mask, BlobBuffers B, int aligned)
{
    __shared__ uint8_t tb[(TY + 2) * TBW];
    __shared__ uint16_t lbase[TY * TX];        // first local slot of a pixel that owns cracks
    __shared__ uint16_t lkey[kTileCracks];     // local slot -> local pixel << 2 | side
    __shared__ uint32_t lparent[kTileCracks];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t gbase;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = blockIdx.y;
    const int W = B.w, H = B.h;
    const size_t N = (size_t)W * H;
    const uint8_t *m = mask + (size_t)img * N;
    uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    const uint32_t ntile = cnts[kCntSegs];
    uint32_t *wl = B.segs + (size_t)img * B.seg_cap;
    uint32_t *cb = B.cbase + (size_t)img * N;
    uint32_t *key = B.ckey + (size_t)img * B.cap, *parent = B.cparent + (size_t)img * B.cap,
             *succ = B.csucc + (size_t)img * B.cap;
    for (uint32_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const uint32_t ent = wl[t];
        const int x0 = (int)(ent & 0x7fffu) * TX, y0 = (int)(ent >> 16) * TY;
        __syncthreads();   // (LDS of the previous tile is no longer read)
        // 1. tile + frame into LDS; outside the image = background
        for (int i = tid; i < (TY + 2) * (TBW / 4); i += 256) {
            const int r = i / (TBW / 4), d = i - r * (TBW / 4);
            const int gy = y0 - 1 + r, gx = x0 - 4 + 4 * d;
            uint32_t v = 0;
            if (gy >= 0 && gy < H) {
                const uint8_t *row = m + (size_t)gy * W;
                if (aligned && gx >= 0 && gx + 4 <= W) {
                    v = *(const uint32_t *)(row + gx);
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (gx + b >= 0 && gx + b < W)
                            v |= (uint32_t)row[gx + b] << (8 * b);
                }
            }
            *(uint32_t *)(tb + r * TBW + 4 * d) = v;
        }
        __syncthreads();
        // 2. this thread's 16 pixels: side bits, crack count
        const int r = tid >> 4, q = tid & 15;
        const uint8_t *c = tb + (r + 1) * TBW + 4 + 16 * q;   // centre row, first pixel
        uint64_t sides64 = 0;
        unsigned cnt = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (c[k] != 0) {
                const unsigned sd = (c[k - TBW] ? 0u : 1u << SN) | (c[k - 1] ? 0u : 1u << SW) |
                                    (c[k + TBW] ? 0u : 1u << SS) | (c[k + 1] ? 0u : 1u << SE);
                sides64 |= (uint64_t)sd << (4 * k);
                cnt += __builtin_popcount(sd);
            }
        }
        // 3. exclusive scan of cnt over the workgroup; one global atomic for the tile
        unsigned inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(inc, o, 64);
            if (lane >= o)
                inc += u;
        }
        if (lane == 63)
            wsum[wave] = inc;
        __syncthreads();
        unsigned woff = 0, total = 0;
#pragma unroll
        for (int wv = 0; wv < 4; wv++) {
            if (wv < wave)
                woff += wsum[wv];
            total += wsum[wv];
        }
        if (tid == 0)
            gbase = total ? atomicAdd(cnts + kCntCracks, total) : 0u;
        __syncthreads();
        const uint32_t base = gbase;
        if (total == 0)
            continue;
        if (base + total > B.cap) {
            if (tid == 0)
                cnts[kCntFlags] = kFlagCrackOverflow;
            continue;
        }
        // 4. slots: local table for the links below, global table for K2 and the later kernels
        const bool big = total > (unsigned)kTileCracks;
        uint32_t loff = woff + inc - cnt;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned sd = (unsigned)(sides64 >> (4 * k)) & 15u;
            if (sd) {
                const int lx = 16 * q + k;
                if (!big)
                    lbase[r * TX + lx] = (uint16_t)loff;
                const uint32_t pix = (uint32_t)(y0 + r) * W + (uint32_t)(x0 + lx);
                cb[pix] = ((base + loff) << 4) | sd;
#pragma unroll
                for (unsigned sdx = 0; sdx < 4; sdx++) {
                    if (sd & (1u << sdx)) {
                        key[base + loff] = pix * 4u + sdx;
                        if (big) {   // every crack its own root, every link K2's
                            parent[base + loff] = base + loff;
                            succ[base + loff] = NIL;
                        } else {
                            lkey[loff] = (uint16_t)(((r * TX + lx) << 2) | sdx);
                            lparent[loff] = loff;
                        }
                        loff++;
                    }
                }
            }
        }
        if (big) {
            if (tid == 0)
                wl[t] = ent | kBigTile;
            continue;
        }
        __syncthreads();
        // 5. links, one crack per thread and pass (a thread's own 16 pixels may all be border
        //    pixels of a horizontal edge: the work is dealt out by slot instead); those that
        //    stay inside the tile are united here
        for (uint32_t li = tid; li < total; li += 256) {
            const unsigned lk = lkey[li];
            const unsigned sdx = lk & 3u;
            const int lx = (int)((lk >> 2) & (TX - 1)), ly = (int)(lk >> 10);   // TX == 256
            const uint8_t *p = tb + (ly + 1) * TBW + 4 + lx;
            Nbr b;
            b.n = p[-TBW] != 0;
            b.s = p[TBW] != 0;
            b.w = p[-1] != 0;
            b.e = p[1] != 0;
            b.nw = p[-TBW - 1] != 0;
            b.ne = p[-TBW + 1] != 0;
            b.sw = p[TBW - 1] != 0;
            b.se = p[TBW + 1] != 0;
            int dx, dy;
            unsigned ns;
            successor(b, sdx, dx, dy, ns);
            const int nx = lx + dx, ny = ly + dy;
            if (nx >= 0 && nx < TX && ny >= 0 && ny < TY) {
                // the successor pixel's side bits, from its own 4-neighbourhood
                const uint8_t *pn = p + dy * TBW + dx;
                const unsigned nsd = (pn[-TBW] ? 0u : 1u << SN) | (pn[-1] ? 0u : 1u << SW) |
                                     (pn[TBW] ? 0u : 1u << SS) | (pn[1] ? 0u : 1u << SE);
                const uint32_t ln = (uint32_t)lbase[ny * TX + nx] + __builtin_popcount(nsd & ((1u << ns) - 1u));
                succ[base + li] = base + ln;
                // unite li and ln in LDS, root = smaller slot (= smaller key: slots follow raster order)
                uint32_t a = li, bb = ln;
                for (;;) {
                    a = lds_find(lparent, a);
                    bb = lds_find(lparent, bb);
                    if (a == bb)
                        break;
                    if (a > bb) {
                        const uint32_t tt = a;
                        a = bb;
                        bb = tt;
                    }
                    if (atomicCAS(lparent + bb, bb, a) == bb)
                        break;
                }
            } else {
                succ[base + li] = NIL;   // leaves the tile: K2
            }
        }
        __syncthreads();
        // 6. the tile's trees, flattened, as the global starting point
        for (uint32_t li = tid; li < total; li += 256)
            parent[base + li] = base + lds_find(lparent, li);
    }
}

// ---- K2: the links that cross a tile border, united with memory-side atomics ---------------------
// Per listed tile: the pixels of its first / last row and column (any pixel with a neighbour
// in another tile); of their cracks, those whose successor lies outside the tile.
__global__ __launch_bounds__(256) void blob_cross_kernel(const uint8_t *__restrict__ mask, BlobBuffers B)
{
    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    const int W = B.w, H = B.h;
    const size_t N = (size_t)W * H;
    const uint8_t *m = mask + (size_t)img * N;
    uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    if (cnts[kCntFlags] != 0)
        return;
    const uint32_t ntile = cnts[kCntSegs];
    const uint32_t *wl = B.segs + (size_t)img * B.seg_cap;
    const uint32_t *cb = B.cbase + (size_t)img * N;
    const uint32_t *key = B.ckey + (size_t)img * B.cap;
    uint32_t *succ = B.csucc + (size_t)img * B.cap, *parent = B.cparent + (size_t)img * B.cap;
    for (uint32_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const uint32_t ent = wl[t];
        const int x0 = (int)(ent & 0x7fffu) * TX, y0 = (int)(ent >> 16) * TY;
        const bool big = (ent & kBigTile) != 0;   // K1 united nothing in this tile: all its pixels, all links
        // border pixels: 256 of row 0, 256 of row 15, 14 + 14 of columns 0 and 255
        for (int i = tid; i < (big ? TX * TY : 2 * TX + 2 * (TY - 2)); i += 256) {
            int lx, ly;
            if (big) {
                lx = i & (TX - 1);
                ly = i >> 8;   // TX == 256
            } else if (i < TX) {
                lx = i;
                ly = 0;
            } else if (i < 2 * TX) {
                lx = i - TX;
                ly = TY - 1;
            } else {
                const int j = i - 2 * TX;
                lx = j < TY - 2 ? 0 : TX - 1;
                ly = 1 + (j < TY - 2 ? j : j - (TY - 2));
            }
            const int x = x0 + lx, y = y0 + ly;
            if (x >= W || y >= H || m[(size_t)y * W + x] == 0)
                continue;
            const Nbr b = load_nbr(m, x, y, W, H);
            const unsigned sd = (b.n ? 0u : 1u << SN) | (b.w ? 0u : 1u << SW) | (b.s ? 0u : 1u << SS) | (b.e ? 0u : 1u << SE);
            if (!sd)
                continue;
            const uint32_t pix = (uint32_t)y * W + x;
            const uint32_t e = cb[pix];
            for (unsigned sdx = 0; sdx < 4; sdx++) {
                if (!(sd & (1u << sdx)))
                    continue;
                int dx, dy;
                unsigned ns;
                successor(b, sdx, dx, dy, ns);
                const int nx = lx + dx, ny = ly + dy;
                if (!big && nx >= 0 && nx < TX && ny >= 0 && ny < TY)
                    continue;   // stayed inside: K1 did it
                const uint32_t ci = slot_of(e, sdx);
                const uint32_t cn = slot_of(cb[(int)pix + dy * W + dx], ns);
                succ[ci] = cn;
                uf_union_pair(parent, key, ci, cn, cnts + kCntFlags);
            }
        }
    }
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
    uint32_t *cslot = B.cslot + (size_t)img * B.cap;
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
            cslot[ci] = slot;   // (the finishing kernel goes from a crack to its border in one step)
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

// (the body returns at several places, always for the whole workgroup; the kernel proper, below, adds
// what has to happen whichever way it went)
__device__ __forceinline__ void blob_finish_body(const uint8_t *__restrict__ mask, const BlobBuffers &B,
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
                   *parent = B.cparent + (size_t)img * B.cap;
    const uint32_t *cslot = B.cslot + (size_t)img * B.cap;
    uint32_t *cpos = B.cpos + (size_t)img * B.cap, *rsel = B.rsel + (size_t)img * B.cap,
             *rnxt = B.rnxt + (size_t)img * B.cap, *rsum = B.rsum + (size_t)img * B.cap;
    BlobAcc *acc = B.acc + (size_t)img * (B.cap / 4);

    // one LDS pool: phases A-C keep the accepted borders in it, phase E the ranking lists
    __shared__ uint32_t pool[3 * kMaxBlobs];
    uint32_t *acc_slot = pool;                   // accepted borders: accumulator slot, by arrival
    uint32_t *acc_key = pool + kMaxBlobs;
    uint32_t *by_rank = pool + 2 * kMaxBlobs;    // accumulator slot by rank
    __shared__ uint32_t n_cand, n_acc, n_sel, flags;
    uint32_t *cand = rsel;                     // (the ranking scratch is free until phase D)
    if (tid == 0) {
        n_cand = n_acc = n_sel = 0;
        flags = 0;
    }
    __syncthreads();

    if (tid == 0)
        cnts[4 + 0] = (uint32_t)wall_clock64();
    // A: outer borders (a00 <= 0: holes run the other way round) with polygon area >= min_area
    //    [REF pipeline.py:562-565: cv.contourArea(contour) >= area_min]
    for (uint32_t s = tid; s < nroots; s += FT) {
        const long long a00 = acc[s].a00;
        if (a00 <= 0 && (double)(-a00) * 0.5 >= min_area)
            cand[atomicAdd(&n_cand, 1u)] = s;   // roots <= cap / 4
    }
    __syncthreads();
    const uint32_t ncand = n_cand;

    if (tid == 0)
        cnts[4 + 1] = (uint32_t)wall_clock64();
    // B: RETR_EXTERNAL = not inside another outer border.  Depth at the start pixel = outer
    //    borders entered minus outer borders left on the way there along its row.
    for (uint32_t j = wave; j < ncand; j += FT / 64) {
        const uint32_t slot = cand[j];
        const uint32_t rk = key[acc[slot].root];
        const int x0 = (int)((rk >> 2) % (uint32_t)W), y0 = (int)((rk >> 2) / (uint32_t)W);
        const uint8_t *row = m + (size_t)y0 * W;
        int d = 0;
        // A border can only lie inside another one through one of ITS holes: unless the
        // start pixel is inside the bounding box of some hole border, the depth is 0 and
        // the row is not walked at all (the usual frame: no holes, or none around a blob).
        int boxed = 0;
        for (uint32_t s2 = lane; s2 < nroots; s2 += 64) {
            const BlobAcc &hb = acc[s2];
            boxed |= hb.a00 > 0 && x0 > hb.xmin && x0 < hb.xmax && y0 > hb.ymin && y0 < hb.ymax;
        }
        if (__builtin_amdgcn_ballot_w64(boxed != 0) == 0) {
            if (lane == 0) {
                const uint32_t k = atomicAdd(&n_acc, 1u);
                if (k < (uint32_t)kMaxBlobs) {
                    acc_slot[k] = slot;
                    acc_key[k] = rk;
                } else {
                    flags = kFlagBlobOverflow;
                }
            }
            continue;
        }
        // eight row positions per lane and pass: their 24 byte loads are independent and in
        // flight together; the dependent look-ups run only where a border edge was met
        for (int xb = 0; xb < x0; xb += 64 * 8) {
            uint8_t cc[8], ll[8], rr[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int x = xb + 64 * k + lane;
                const bool inb = x < x0;
                cc[k] = inb ? row[x] : (uint8_t)0;
                ll[k] = inb && x > 0 ? row[x - 1] : (uint8_t)0;
                rr[k] = inb ? row[x + 1] : (uint8_t)1;   // x + 1 <= x0 < W
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (cc[k] == 0 || (ll[k] != 0 && rr[k] != 0))
                    continue;
                const int x = xb + 64 * k + lane;
                const uint32_t e = cb[(size_t)y0 * W + x];
                if (ll[k] == 0 && acc[cslot[slot_of(e, SW)]].a00 <= 0)
                    d++;
                if (rr[k] == 0 && acc[cslot[slot_of(e, SE)]].a00 <= 0)
                    d--;
            }
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

    if (tid == 0)
        cnts[4 + 2] = (uint32_t)wall_clock64();
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

    if (tid == 0)
        cnts[4 + 3] = (uint32_t)wall_clock64();
    // D: the cracks of accepted blobs, compacted
#pragma unroll 4
    for (uint32_t c0 = 0; c0 < nc; c0 += FT) {
        const uint32_t ci = c0 + tid;
        if (ci < nc && acc[cslot[ci]].out >= 0) {
            const uint32_t pos = atomicAdd(&n_sel, 1u);
            rsel[pos] = ci;
            cpos[ci] = pos;
        }
    }
    __syncthreads();
    const uint32_t nsel = n_sel;

    if (tid == 0)
        cnts[4 + 4] = (uint32_t)wall_clock64();
    // E: list ranking along each cycle, cut in front of its root crack.  rsum[j] ends as the
    //    number of pixel changes from crack j to the end of its cycle, so that crack j
    //    belongs to point (transitions of the cycle - rsum[j]).
#pragma unroll 4
    for (uint32_t j = tid; j < nsel; j += FT) {
        const uint32_t ci = rsel[j], cn = succ[ci];
        rsum[j] = (key[ci] >> 2) != (key[cn] >> 2) ? 1u : 0u;
        rnxt[j] = parent[cn] == cn ? NIL : cpos[cn];
    }
    __syncthreads();
    if (tid == 0)
        cnts[4 + 5] = (uint32_t)wall_clock64();
    // pointer jumping, double-buffered: every step reads (sa, na) and writes (sb, nb).
    // Lists of up to kLdsRank cracks (a frame's handful of people-sized blobs) are ranked
    // in LDS as 16-bit entries; longer ones in the global scratch.
    constexpr uint32_t kLdsRank = 3 * kMaxBlobs;   // one packed dword (sum << 16 | next) per crack
    constexpr int PER = 3 * kMaxBlobs / FT;         // list elements per thread
    constexpr uint32_t NIL16 = 0xffffu;
    uint32_t *sa = rsum, *na = rnxt, *sb = B.rsum2 + (size_t)img * B.cap, *nb = B.rnxt2 + (size_t)img * B.cap;
    const bool in_lds = nsel <= kLdsRank && nsel < NIL16;
    int go = 1;
    if (in_lds) {
        for (uint32_t j = tid; j < nsel; j += FT)
            pool[j] = (sa[j] << 16) | (na[j] == NIL ? NIL16 : na[j]);
        __syncthreads();
        for (int step = 0; step < 18 && go; step++) {
            // every thread reads what its elements need, then all write: one buffer
            uint32_t nv[PER];
            int any = 0;
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const uint32_t j = (uint32_t)q * FT + tid;
                nv[q] = 0;
                if (j < nsel) {
                    const uint32_t v = pool[j], n = v & NIL16;
                    nv[q] = v;
                    if (n != NIL16) {
                        const uint32_t u = pool[n];
                        nv[q] = (((v >> 16) + (u >> 16)) << 16) | (u & NIL16);
                        any = 1;
                    }
                }
            }
            go = __syncthreads_or(any);
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const uint32_t j = (uint32_t)q * FT + tid;
                if (j < nsel)
                    pool[j] = nv[q];
            }
            __syncthreads();
        }
        for (uint32_t j = tid; j < nsel; j += FT)
            sa[j] = pool[j] >> 16;
        __syncthreads();
    } else {
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
    }
    if (go) {   // a list without an end: cannot happen (every cycle holds its root crack)
        if (tid == 0)
            hdr->flags |= kFlagInternal;
        return;
    }

    if (tid == 0)
        cnts[4 + 6] = (uint32_t)wall_clock64();
    // F: every crack writes its pixel into its point's place (the cracks of one run write
    //    the same value); the run that closes the cycle on the start pixel is point 0 again
#pragma unroll 4
    for (uint32_t j = tid; j < nsel; j += FT) {
        const uint32_t ci = rsel[j];
        const BlobAcc *a = acc + cslot[ci];
        const int tot = a->trans;
        const int idx = tot - (int)sa[j];
        const int np = tot > 0 ? tot : 1;
        if (idx < np) {
            const uint32_t p = key[ci] >> 2;
            points[(uint32_t)recs[a->out].point_offset + (uint32_t)idx] = (p % (uint32_t)W) | ((p / (uint32_t)W) << 16);
        }
    }
    __syncthreads();
    if (tid == 0)
        cnts[4 + 7] = (uint32_t)wall_clock64();
}

__global__ __launch_bounds__(FT) void blob_finish_kernel(const uint8_t *__restrict__ mask, BlobBuffers B,
                                                         double min_area)
{
    blob_finish_body(mask, B, min_area);
    // Behind the body, whichever way it left: (1) the head of the output block -- header, records,
    // points, as far as the mirror reaches -- goes to the host-visible mirror (plain stores over the
    // link, visible to the host when the kernel has completed: no copy operation queued behind the
    // extraction); (2) the counters are cleared for the next extraction (no memset in front of it),
    // a copy kept for diagnostics.
    const int img = blockIdx.x, tid = threadIdx.x;
    uint32_t *cnts = B.counters + (size_t)img * kBlobCounters;
    const uint32_t *out = B.out + (size_t)img * B.out_words;
    __syncthreads();
    if (B.host_out != nullptr) {
        const BlobHeader *hdr = (const BlobHeader *)out;
        unsigned long long words = kBlobHeaderWords;
        if (hdr->flags == 0)
            words += (unsigned long long)hdr->n_blobs * kBlobRecordWords + hdr->n_points;
        if (words > B.host_words)
            words = B.host_words;
        uint32_t *dst = B.host_out + (size_t)img * B.host_words;
        for (uint32_t i = tid; i < (uint32_t)words; i += FT)
            dst[i] = out[i];
    }
    if (tid < 4) {
        cnts[kCntLast + tid] = cnts[tid];
        cnts[tid] = 0;
    }
}

}  // namespace

hipError_t blob_extract(const uint8_t *mask, const BlobBuffers &B, double min_area, hipStream_t s)
{
    if (B.n_img < 1 || B.h < 1 || B.w < 1 || B.h > 65535 || B.w > 65535 || B.n_img > 65535 ||
        (B.h + 3) / 4 > 65535 || B.cap < 16 || B.cap > (1u << 28) || (B.cap & 3u) ||
        B.out_words < kBlobHeaderWords || (size_t)B.h * B.w > (1u << 30))
        return hipErrorInvalidValue;
    const unsigned tiles_x = (unsigned)((B.w + TX - 1) / TX), tiles_y = (unsigned)((B.h + TY - 1) / TY);
    if (B.seg_cap < tiles_x * tiles_y)
        return hipErrorInvalidValue;
    const dim3 g(tiles_x, (tiles_y + 3) / 4, (unsigned)B.n_img);
    const int aligned = B.w % 4 == 0 && (uintptr_t)mask % 4 == 0;
    hipLaunchKernelGGL(blob_flag_kernel, g, dim3(256), 0, s, mask, B, aligned);
    // the tile and crack counts live on the device: fixed modest grids, grid-stride loops
    const unsigned per_img = B.n_img >= 16 ? 64u : 256u;
    const dim3 gt(per_img < tiles_x * tiles_y ? per_img : tiles_x * tiles_y, (unsigned)B.n_img);
    hipLaunchKernelGGL(blob_tile_kernel, gt, dim3(256), 0, s, mask, B, aligned);
    hipLaunchKernelGGL(blob_cross_kernel, gt, dim3(256), 0, s, mask, B);
    const dim3 g2(64, (unsigned)B.n_img);
    hipLaunchKernelGGL(blob_roots_kernel, g2, dim3(256), 0, s, B);
    hipLaunchKernelGGL(blob_stats_kernel, g2, dim3(256), 0, s, B);
    hipLaunchKernelGGL(blob_finish_kernel, dim3((unsigned)B.n_img), dim3(FT), 0, s, mask, B, min_area);
    return hipGetLastError();
}

}  // namespace gofk
