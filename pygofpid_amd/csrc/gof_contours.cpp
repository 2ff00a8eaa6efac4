// Host-side contour retrieval for GOFPID.detect step 4 (the consumer of the
// foreground mask): cv.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_NONE) as
// called by the reference's helpers.find_contours (pygofpid/helpers.py:125-150).
//
// This is NOT on the GPU hot path (SURVEY.md section 8(f) lists a GPU
// connected-component version as "next #1"); it exists so that the drop-in
// GOFPID.detect() can run end to end on a box without cv2.  It follows Suzuki &
// Abe's border-following algorithm (the published algorithm behind
// cv.findContours, [OPENCV-RECALL]): raster scan, outer/hole border starts,
// 8-connected foreground, parent bookkeeping; only outer borders whose parent
// is the frame are returned (RETR_EXTERNAL), every border pixel is kept
// (CHAIN_APPROX_NONE).  PARITY UNPINNED against cv2 (point order of degenerate
// borders is the part most likely to differ; area / moments / bounding boxes,
// which is all the reference consumes, do not depend on it).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#include "gofpid_hip.h"

namespace {

// 8-neighbourhood in clockwise order starting at "east" (image coords, y down)
const int kDy[8] = {0, 1, 1, 1, 0, -1, -1, -1};
const int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};

inline int dir_of(int dy, int dx)
{
    for (int d = 0; d < 8; d++)
        if (kDy[d] == dy && kDx[d] == dx)
            return d;
    return 0;
}

}  // namespace

extern "C" int gof_find_external_contours(const uint8_t *mask, int h, int w, int32_t **points_xy,
                                          int32_t **offsets, int *n_contours)
{
    if (!mask || !points_xy || !offsets || !n_contours || h < 1 || w < 1)
        return GOF_EINVAL;
    *points_xy = nullptr;
    *offsets = nullptr;
    *n_contours = 0;
    const int W = w + 2, H = h + 2;   // one-pixel frame of zeros, as cv2 assumes
    // A foreground mask is mostly zeros.  The label image is a per-thread buffer
    // that is all zero between calls (only the labels of non-zero mask bytes are
    // ever written, and they are cleared again on the way out), only non-zero mask
    // bytes are transcribed, and both passes skip zeros eight bytes at a time on
    // the mask itself: a label is non-zero exactly where the mask is, whatever sign
    // or magnitude the border following gives it.
    static thread_local std::vector<int32_t> label_buf;
    if (label_buf.size() < (size_t)W * H) {
        try {
            label_buf.assign((size_t)W * H, 0);
        } catch (const std::bad_alloc &) {
            return GOF_ENOMEM;
        }
    }
    int32_t *const f = label_buf.data();
    std::vector<uint8_t> row_any((size_t)h, 0);
    struct Cleaner {   // labels back to zero, whichever way the function is left
        const uint8_t *mask;
        int32_t *f;
        const uint8_t *row_any;
        int h, w, W;
        ~Cleaner()
        {
            for (int y = 0; y < h; y++) {
                if (!row_any[y])
                    continue;
                const uint8_t *r = mask + (size_t)y * w;
                int32_t *fr = f + (size_t)(y + 1) * W + 1;
                int x = 0;
                for (; x + 8 <= w; x += 8) {
                    uint64_t v;
                    memcpy(&v, r + x, 8);
                    if (!v)
                        continue;
                    for (int k = 0; k < 8; k++)
                        fr[x + k] = 0;
                }
                for (; x < w; x++)
                    fr[x] = 0;
            }
        }
    } cleaner{mask, f, row_any.data(), h, w, W};
    for (int y = 0; y < h; y++) {
        const uint8_t *r = mask + (size_t)y * w;
        int32_t *fr = f + (size_t)(y + 1) * W + 1;
        bool any = false;
        int x = 0;
        for (; x + 8 <= w; x += 8) {
            if (x + 32 <= w) {   // thirty-two zeros at once
                uint64_t q[4];
                memcpy(q, r + x, 32);
                if (!(q[0] | q[1] | q[2] | q[3])) {
                    x += 24;
                    continue;
                }
            }
            uint64_t v;
            memcpy(&v, r + x, 8);
            if (!v)
                continue;
            any = true;
            for (int k = 0; k < 8; k++)
                if (r[x + k])
                    fr[x + k] = 1;
        }
        for (; x < w; x++)
            if (r[x]) {
                any = true;
                fr[x] = 1;
            }
        row_any[(size_t)y] = any;
    }

    std::vector<int32_t> pts;             // x, y pairs of every returned contour
    std::vector<int32_t> offs(1, 0);
    std::vector<int32_t> parent(2, 0);    // border id -> parent id (1 = frame)
    std::vector<uint8_t> is_hole(2, 0);
    is_hole[1] = 1;                       // the frame behaves like a hole border
    int nbd = 1;

    for (int i = 1; i < H - 1; i++) {
        if (!row_any[(size_t)(i - 1)])
            continue;
        int lnbd = 1;
        const uint8_t *r = mask + (size_t)(i - 1) * w;
        for (int j = 1; j < W - 1; j++) {
            const int x = j - 1;
            if ((x & 7) == 0 && x + 8 <= w) {   // eight zeros at once
                uint64_t v;
                memcpy(&v, r + x, 8);
                if (!v) {
                    j += 7;
                    continue;
                }
            }
            if (r[x] == 0)
                continue;
            const int32_t fij = f[(size_t)i * W + j];
            int i2 = 0, j2 = 0;
            bool outer = false, start = false;
            if (fij == 1 && f[(size_t)i * W + j - 1] == 0) {
                outer = true;
                start = true;
                i2 = i;
                j2 = j - 1;
            } else if (fij >= 1 && f[(size_t)i * W + j + 1] == 0) {
                start = true;
                i2 = i;
                j2 = j + 1;
                if (fij > 1)
                    lnbd = fij;
            }
            if (start) {
                ++nbd;
                // parent per Suzuki & Abe, table 1
                const int bprime = lnbd;
                int par;
                if (outer)
                    par = is_hole[(size_t)bprime] ? bprime : parent[(size_t)bprime];
                else
                    par = is_hole[(size_t)bprime] ? parent[(size_t)bprime] : bprime;
                parent.push_back(par);
                is_hole.push_back(outer ? 0 : 1);
                const bool keep = outer && par == 1;
                const size_t first_pt = pts.size();

                // (3.1) clockwise search around (i, j) starting from (i2, j2)
                int d0 = dir_of(i2 - i, j2 - j);
                int i1 = -1, j1 = -1;
                for (int s = 0; s < 8; s++) {
                    const int d = (d0 + s) & 7;
                    const int yy = i + kDy[d], xx = j + kDx[d];
                    if (f[(size_t)yy * W + xx] != 0) {
                        i1 = yy;
                        j1 = xx;
                        break;
                    }
                }
                if (i1 < 0) {
                    f[(size_t)i * W + j] = -nbd;   // isolated pixel
                    if (keep) {
                        pts.push_back(j - 1);
                        pts.push_back(i - 1);
                    }
                } else {
                    i2 = i1;
                    j2 = j1;
                    int i3 = i, j3 = j;
                    for (;;) {
                        // (3.3) counter-clockwise search around (i3, j3), starting after (i2, j2)
                        const int dstart = dir_of(i2 - i3, j2 - j3);
                        int i4 = -1, j4 = -1;
                        bool east_zero_examined = false;
                        for (int s = 1; s <= 8; s++) {
                            const int d = (dstart - s) & 7;
                            const int yy = i3 + kDy[d], xx = j3 + kDx[d];
                            if (f[(size_t)yy * W + xx] != 0) {
                                i4 = yy;
                                j4 = xx;
                                break;
                            }
                            if (d == 0)
                                east_zero_examined = true;
                        }
                        // (3.4)
                        int32_t &cur = f[(size_t)i3 * W + j3];
                        if (east_zero_examined)
                            cur = -nbd;
                        else if (cur == 1)
                            cur = nbd;
                        if (keep) {
                            pts.push_back(j3 - 1);
                            pts.push_back(i3 - 1);
                        }
                        // (3.5)
                        if (i4 == i && j4 == j && i3 == i1 && j3 == j1)
                            break;
                        i2 = i3;
                        j2 = j3;
                        i3 = i4;
                        j3 = j4;
                    }
                }
                if (keep) {
                    (void)first_pt;
                    offs.push_back((int32_t)(pts.size() / 2));
                }
            }
            // (4)
            const int32_t now = f[(size_t)i * W + j];
            if (now != 1)
                lnbd = now < 0 ? -now : now;
        }
    }

    const int n = (int)offs.size() - 1;
    int32_t *p = (int32_t *)malloc((pts.size() + 2) * sizeof(int32_t));
    int32_t *o = (int32_t *)malloc(offs.size() * sizeof(int32_t));
    if (!p || !o) {
        free(p);
        free(o);
        return GOF_ENOMEM;
    }
    if (!pts.empty())
        memcpy(p, pts.data(), pts.size() * sizeof(int32_t));
    memcpy(o, offs.data(), offs.size() * sizeof(int32_t));
    *points_xy = p;
    *offsets = o;
    *n_contours = n;
    return GOF_OK;
}

extern "C" void gof_free(void *p) { free(p); }
