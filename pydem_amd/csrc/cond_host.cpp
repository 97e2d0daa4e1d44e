// cond_host.cpp -- host-side inner loops of the elevation conditioning (no device code).
//
// The reference conditions a tile before the slope stencil (pydem/dem_processing.py:
// calc_fill_pit_artifacts :396-426, calc_fill_flats :551-579 with _fill_flat :308-394,
// calc_pit_drain_paths :428-548; helpers pydem/utils.py:342-468).  Those are region-by-region /
// pit-by-pit sequential algorithms; pydem_amd/conditioning.py keeps their vectorised prologues in
// numpy/scipy (3x3 filters, connected-component labels, the argsort whose tie order is part of the
// result) and hands the per-region / per-pit loops -- 20 k scipy.ndimage calls on tiny windows for
// a 768^2 tile -- to the three functions below.  Every floating-point expression keeps numpy's
// operation order (this file is compiled with -ffp-contract=off like the kernels); results are pinned
// bit for bit by tests/golden/g5_* and g7_* and by tests that compare with the pure-numpy versions.
#include "internal.h"
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>

namespace {

struct Box { int64_t r0, r1, c0, c1; };     // half-open bounding box of a label

// scipy.ndimage.find_objects: bounding boxes by label (labels are 1..nlab; absent labels keep r0 > r1)
void find_boxes(const int32_t *lab, int64_t n, int64_t m, int32_t nlab, std::vector<Box> &box)
{
    box.assign((size_t)nlab + 1, Box{n, 0, m, 0});
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < m; j++) {
            const int32_t k = lab[i * m + j];
            if (k <= 0) continue;
            Box &b = box[(size_t)k];
            if (i < b.r0) b.r0 = i;
            if (i + 1 > b.r1) b.r1 = i + 1;
            if (j < b.c0) b.c0 = j;
            if (j + 1 > b.c1) b.c1 = j + 1;
        }
}

// numpy pairwise sum of a contiguous float64 vector (np.add.reduce), see oracle/pydem_oracle.c
double np_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

// window helpers ---------------------------------------------------------------------------------
struct Win {
    int64_t h, w;
    std::vector<uint8_t> region, ring, source, drain, pinned, tmp;
    std::vector<double> dh, dl, nd;
};

// 8-connected dilation inside the window (scipy binary_dilation, border_value 0)
void dilate8(const std::vector<uint8_t> &a, int64_t h, int64_t w, std::vector<uint8_t> &out)
{
    out.assign((size_t)(h * w), 0);
    for (int64_t i = 0; i < h; i++)
        for (int64_t j = 0; j < w; j++) {
            if (!a[(size_t)(i * w + j)]) continue;
            for (int64_t ii = (i > 0 ? i - 1 : 0); ii <= (i + 1 < h ? i + 1 : h - 1); ii++)
                for (int64_t jj = (j > 0 ? j - 1 : 0); jj <= (j + 1 < w ? j + 1 : w - 1); jj++) out[(size_t)(ii * w + jj)] = 1;
        }
}

// utils.get_border_mask (:342-370) with its shortcut: when the window interior is entirely region,
// everything outside the region counts as border
void neighbour_ring(const std::vector<uint8_t> &mask, int64_t h, int64_t w, std::vector<uint8_t> &ring, std::vector<uint8_t> &tmp)
{
    bool all = true, any = false;
    for (int64_t i = 1; i + 1 < h; i++)
        for (int64_t j = 1; j + 1 < w; j++) {
            if (mask[(size_t)(i * w + j)]) any = true; else all = false;
        }
    ring.assign((size_t)(h * w), 0);
    if (all && any) {
        for (size_t k = 0; k < ring.size(); k++) ring[k] = !mask[k];
        return;
    }
    dilate8(mask, h, w, tmp);
    for (size_t k = 0; k < ring.size(); k++) ring[k] = tmp[k] && !mask[k];
}

// utils.get_distance (:374-402): within-region (1, sqrt 2) chamfer distance by Jacobi sweeps over the whole
// window (3x3 / cross minimum filters with scipy's default 'reflect' boundary = in-window neighbours),
// stopping as soon as every region cell has SOME finite value -- not at convergence
void chamfer(const std::vector<uint8_t> &region, const std::vector<uint8_t> &seeds, int64_t h, int64_t w, std::vector<double> &d,
             std::vector<double> &nd)
{
    const double big = (double)(h * w), sqrt2 = sqrt(2.0);
    d.assign((size_t)(h * w), big);
    for (size_t k = 0; k < d.size(); k++) if (seeds[k]) d[k] = 0.0;
    nd.resize(d.size());
    for (int64_t it = 0; it < h * w; it++) {
        for (int64_t i = 0; i < h; i++)
            for (int64_t j = 0; j < w; j++) {
                const size_t c = (size_t)(i * w + j);
                if (!region[c]) continue;
                double cross = d[c], full = d[c];
                const int64_t i0 = i > 0 ? i - 1 : 0, i1 = i + 1 < h ? i + 1 : h - 1, j0 = j > 0 ? j - 1 : 0, j1 = j + 1 < w ? j + 1 : w - 1;
                for (int64_t ii = i0; ii <= i1; ii++)
                    for (int64_t jj = j0; jj <= j1; jj++) {
                        const double v = d[(size_t)(ii * w + jj)];
                        if (v < full) full = v;
                        if ((ii == i || jj == j) && v < cross) cross = v;
                    }
                const double a = cross + 1, b = full + sqrt2;
                const double best = a < b ? a : b;
                nd[c] = best < d[c] ? best : d[c];
            }
        bool done = true;
        for (size_t c = 0; c < d.size(); c++)
            if (region[c]) { d[c] = nd[c]; if (!(d[c] < big)) done = false; }
        if (done) break;
    }
}

// utils.find_centroid (:450-468): region cell nearest to the centre of mass (first one in raster order on ties)
size_t centre_cell(const std::vector<uint8_t> &region, int64_t h, int64_t w)
{
    double si = 0, sj = 0, cnt = 0;
    for (int64_t i = 0; i < h; i++)
        for (int64_t j = 0; j < w; j++)
            if (region[(size_t)(i * w + j)]) { si += (double)i; sj += (double)j; cnt += 1; }
    const double cy = si / cnt, cx = sj / cnt;
    size_t best = 0;
    double bd = INFINITY;
    for (int64_t i = 0; i < h; i++)
        for (int64_t j = 0; j < w; j++) {
            const size_t c = (size_t)(i * w + j);
            if (!region[c]) continue;
            const double dy = (double)i - cy, dx = (double)j - cx;
            const double dist = sqrt(dy * dy + dx * dx);
            if (dist < bd) { bd = dist; best = c; }
        }
    return best;
}

}  // namespace

extern "C" {

// calc_fill_pit_artifacts (:396-426) for the labelled candidate regions: raise[c] = 1 where the cell must be
// lifted by one unit.  `elev` is the surface as float64 (exact for integer and float32 inputs; f32: it holds float32 values).
int pydem_cond_pit_artifacts(const double *elev, int64_t n, int64_t m, const int32_t *lab, int32_t nlab, double max_area,
                             int f32, uint8_t *raise)
{
    std::vector<Box> box;
    find_boxes(lab, n, m, nlab, box);
    memset(raise, 0, (size_t)(n * m));
    std::vector<uint8_t> body, grown;
    for (int32_t k = 1; k <= nlab; k++) {
        const Box b = box[(size_t)k];
        if (b.r0 >= b.r1) continue;
        if (b.r0 == 0 || b.c0 == 0 || b.r1 == n || b.c1 == m) continue;          // the one-pixel rim must lie inside the array (:414-415)
        const int64_t R0 = b.r0 - 1, C0 = b.c0 - 1, h = b.r1 - b.r0 + 2, w = b.c1 - b.c0 + 2;
        body.assign((size_t)(h * w), 0);
        int64_t size = 0;
        double level = 0;
        bool have = false;
        for (int64_t i = 0; i < h; i++)
            for (int64_t j = 0; j < w; j++)
                if (lab[(R0 + i) * m + C0 + j] == k) {
                    body[(size_t)(i * w + j)] = 1; size++;
                    if (!have) { level = elev[(R0 + i) * m + C0 + j]; have = true; }
                }
        if ((double)size > max_area) continue;
        dilate8(body, h, w, grown);                                                 // maximum_filter(body, 3x3), reflect == in-window
        bool ok = true;
        for (int64_t i = 0; i < h && ok; i++)
            for (int64_t j = 0; j < w; j++) {
                const size_t c = (size_t)(i * w + j);
                if (!grown[c] || body[c]) continue;
                const double v = elev[(R0 + i) * m + C0 + j];
                // (`rim - 1` is computed in the array's dtype: a float32 surface rounds it to float32, :424)
                if (!(f32 ? ((float)v - 1.0f == (float)level) : (v - 1 == level))) { ok = false; break; }
            }
        if (!ok) continue;
        for (int64_t i = 0; i < h; i++)
            for (int64_t j = 0; j < w; j++)
                if (body[(size_t)(i * w + j)]) raise[(R0 + i) * m + C0 + j] = 1;
    }
    return 0;
}

// _fill_flat (:308-394) for every labelled flat of calc_fill_flats (:551-579).  `data` is the unmodified
// surface, `built` (a copy of it on entry) receives the re-surfaced flats.
int pydem_cond_fill_flats(const double *data, double *built, int64_t n, int64_t m, const int32_t *lab, int32_t nlab,
                          double source_tol, int peaks, int pits)
{
    std::vector<Box> box;
    find_boxes(lab, n, m, nlab, box);
    Win W;
    std::vector<double> roi;
    std::vector<uint8_t> edge;
    for (int32_t k = 1; k <= nlab; k++) {
        const Box b = box[(size_t)k];
        if (b.r0 >= b.r1) continue;
        const int64_t R0 = b.r0 > 0 ? b.r0 - 1 : 0, R1 = b.r1 + 1 < n ? b.r1 + 1 : n;
        const int64_t C0 = b.c0 > 0 ? b.c0 - 1 : 0, C1 = b.c1 + 1 < m ? b.c1 + 1 : m;
        const int64_t h = R1 - R0, w = C1 - C0;
        const size_t sz = (size_t)(h * w);
        roi.resize(sz); edge.resize(sz); W.region.assign(sz, 0);
        int64_t count = 0;
        double level = 0;
        bool have = false;
        for (int64_t i = 0; i < h; i++)
            for (int64_t j = 0; j < w; j++) {
                const int64_t gi = R0 + i, gj = C0 + j;
                const size_t c = (size_t)(i * w + j);
                roi[c] = data[gi * m + gj];
                edge[c] = gi == 0 || gi == n - 1 || gj == 0 || gj == m - 1;
                if (lab[gi * m + gj] == k) {
                    W.region[c] = 1; count++;
                    if (!have) { level = roi[c]; have = true; }
                }
            }
        auto out = [&](size_t c) -> double & { return built[(R0 + (int64_t)(c / (size_t)w)) * m + C0 + (int64_t)(c % (size_t)w)]; };
        if (sz <= 9 && count == 1) {                                        // single pixel in a tiny window (:312-325)
            int64_t n_high = 0;
            double mn = INFINITY;
            for (size_t c = 0; c < sz; c++) if (roi[c] > level) { n_high++; if (roi[c] < mn) mn = roi[c]; }
            size_t rc = 0;
            for (size_t c = 0; c < sz; c++) if (W.region[c]) rc = c;
            if (n_high == (int64_t)sz - 1) continue;                         // a true pit: leave it
            if (n_high > 0) { const double d = mn - level; out(rc) += (1.0 < d ? 1.0 : d) - 0.01; }
            else if (peaks) out(rc) += 0.5;
            continue;
        }
        neighbour_ring(W.region, h, w, W.ring, W.tmp);
        W.drain.assign(sz, 0); W.source.assign(sz, 0);
        bool any_source = false, any_drain = false;
        for (size_t c = 0; c < sz; c++) {
            if (!W.ring[c]) continue;
            if (roi[c] == level) { W.drain[c] = 1; any_drain = true; }
            if (roi[c] > level) { W.source[c] = 1; any_source = true; }
        }
        const std::vector<uint8_t> *pinned = nullptr;                        // cells whose value is set, not interpolated
        double top = 0;
        if (any_source) {                                                    // gentle uphill rim (:343-347)
            double lowest = INFINITY;
            for (size_t c = 0; c < sz; c++) if (W.source[c] && roi[c] < lowest) lowest = roi[c];
            top = level + 1.0 < lowest ? level + 1.0 : lowest;
            for (size_t c = 0; c < sz; c++) if (W.source[c] && !(roi[c] <= lowest + source_tol)) W.source[c] = 0;
        } else if (peaks) {                                                  // summit plateau: drain away from its centre (:348-354)
            top = level + 0.5;
            const size_t ci = centre_cell(W.region, h, w);
            out(ci) = top;
            W.source[ci] = 1;
            pinned = &W.source;
        } else continue;
        if (any_drain) {
        } else {
            bool on_edge = false;
            for (size_t c = 0; c < sz; c++) if (W.region[c] && edge[c]) on_edge = true;
            if (on_edge) {                                                   // river bed leaving through the tile edge (:362-366)
                bool rest = false;
                for (size_t c = 0; c < sz; c++) { W.drain[c] = W.region[c] && edge[c]; if (W.region[c] && !W.drain[c]) rest = true; }
                pinned = &W.drain;
                if (!rest) continue;
            } else if (pits) {                                               // closed depression: drain towards its centre (:367-371)
                const size_t ci = centre_cell(W.region, h, w);
                W.drain[ci] = 1;
                pinned = &W.drain;
            } else continue;
        }
        chamfer(W.region, W.source, h, w, W.dh, W.nd);
        chamfer(W.region, W.drain, h, w, W.dl, W.nd);
        for (size_t c = 0; c < sz; c++) {
            if (!W.region[c] || (pinned && (*pinned)[c])) continue;
            const double dl2 = W.dl[c] * W.dl[c], dh2 = W.dh[c] * W.dh[c];
            out(c) = (top * dl2 + level * dh2) / (dl2 + dh2);
        }
    }
    return 0;
}

// calc_pit_drain_paths (:428-548): pits in the given order (the caller's numpy argsort), the surface `e` is
// edited in place and sequentially, later pits see earlier paths.
int pydem_cond_pit_paths(double *e, int64_t n, int64_t m, const int64_t *pits, int64_t npits, const double *dX, int64_t ndX,
                         const double *dY, int max_iter, int max_dist, double max_dist_XY, int dtype_mode, int64_t *n_failed,
                         int64_t *iter_used)
{
    // dtype_mode: the reference edits the array in ITS dtype (:535-539).  0: float64.  1: an integer surface -- the path
    // values are truncated towards zero when they are stored.  2: a float32 surface -- the drop is a float32 difference and
    // the stored values round to float32.  `e` holds the values as float64 in every mode.
    const int64_t NN = n * m;
    std::vector<int32_t> stamp((size_t)NN, -1);      // 2*p: in area of pit p, 2*p+1: on its rim
    std::vector<int64_t> rim, trail, fresh, chain, outlet, keep;
    std::vector<double> reach;
    int64_t failed = 0, used = 0;
    auto add_ring = [&](int64_t c, int32_t p) {
        const int64_t i = c / m, j = c - i * m;
        for (int64_t ii = (i > 0 ? i - 1 : 0); ii <= (i + 1 < n ? i + 1 : n - 1); ii++)
            for (int64_t jj = (j > 0 ? j - 1 : 0); jj <= (j + 1 < m ? j + 1 : m - 1); jj++) {
                const int64_t t = ii * m + jj;
                if (t == c) continue;
                if (stamp[(size_t)t] == 2 * p || stamp[(size_t)t] == 2 * p + 1) continue;
                stamp[(size_t)t] = 2 * p + 1;
                rim.push_back(t);
            }
    };
    for (int64_t q = 0; q < npits; q++) {
        const int32_t p = (int32_t)q;
        const int64_t pit = pits[q];
        rim.clear(); trail.clear(); outlet.clear();
        trail.push_back(pit);
        stamp[(size_t)pit] = 2 * p;
        const double floor_ = e[pit];
        add_ring(pit, p);
        bool found = false;
        int it = 0;
        for (it = 0; it < max_iter; it++) {
            if (rim.empty()) break;
            double lowest = INFINITY;
            bool has_nan = false;
            for (int64_t t : rim) { if (e[t] < lowest) lowest = e[t]; if (isnan(e[t])) has_nan = true; }
            if (has_nan) break;       // np.min propagates NaN: nothing equals it, the region stops growing and the pit fails
            fresh.clear();
            for (int64_t t : rim) if (e[t] == lowest) fresh.push_back(t);
            std::sort(fresh.begin(), fresh.end());
            if (lowest < floor_) { outlet = fresh; found = true; break; }
            // area += rim cells at the lowest height; rim loses them and gains their unseen neighbours
            size_t wr = 0;
            for (size_t k = 0; k < rim.size(); k++) if (!(e[rim[k]] == lowest)) rim[wr++] = rim[k];
            rim.resize(wr);
            for (int64_t t : fresh) { trail.push_back(t); stamp[(size_t)t] = 2 * p; }
            for (int64_t t : fresh) add_ring(t, p);
        }
        if (!found) { failed++; continue; }
        if (it + 1 > used) used = it + 1;
        const int64_t ip = pit / m, jp = pit - ip * m;
        if (max_dist) {                                                       // index-space reach (:485-493)
            keep.clear();
            for (int64_t t : outlet) {
                const int64_t oi = t / m, oj = t - oi * m;
                const double di = (double)(ip - oi), dj = (double)(jp - oj);
                if (sqrt(di * di + dj * dj) <= (double)max_dist) keep.push_back(t);
            }
            if (keep.empty()) { failed++; continue; }
            outlet = keep;
        }
        reach.resize(outlet.size());
        for (size_t k = 0; k < outlet.size(); k++) {
            const int64_t oi = outlet[k] / m, oj = outlet[k] - oi * m;
            const int64_t lo = ip < oi ? ip : oi, hi = ip < oi ? oi : ip;
            double dxm;
            if (ip == oi) dxm = dX[ip < ndX - 1 ? ip : ndX - 1];               // _get_dX_mean :1993-1997
            else dxm = np_sum(dX + lo, hi - lo) / (double)(hi - lo);
            const double run = dxm * (double)(jp - oj);
            const double rise = np_sum(dY + lo, hi - lo);
            reach[k] = sqrt(run * run + rise * rise);
        }
        if (max_dist_XY != 0 && !isnan(max_dist_XY)) {                        // metric reach (:502-508)
            keep.clear();
            std::vector<double> r2;
            for (size_t k = 0; k < outlet.size(); k++) if (reach[k] <= max_dist_XY) { keep.push_back(outlet[k]); r2.push_back(reach[k]); }
            if (keep.empty()) { failed++; continue; }
            outlet = keep; reach = r2;
        }
        int64_t end = outlet[0];
        if (outlet.size() > 1) {
            double mn = reach[0];
            for (double v : reach) if (v < mn) mn = v;
            for (size_t k = 0; k < outlet.size(); k++) if (reach[k] == mn) { end = outlet[k]; break; }
        }
        // prune the trail, walking back from the outlet, to an 8-connected chain (:516-532)
        chain = trail;
        chain.push_back(end);
        int64_t k = (int64_t)chain.size() - 2;
        while (k > 0) {
            const int64_t a = chain[(size_t)k], b = chain[(size_t)k + 1];
            const int64_t ai = a / m, aj = a - ai * m, bi = b / m, bj = b - bi * m;
            if (llabs(ai - bi) <= 1 && llabs(aj - bj) <= 1) k -= 1;
            else {
                chain.erase(chain.begin() + k);
                if (k > (int64_t)chain.size() - 2) k = (int64_t)chain.size() - 2;
            }
            if (chain[(size_t)k] == pit) break;
        }
        // elevations fall linearly along the chain (:535-539)
        if (e[pit] < e[end]) {
            double mn = INFINITY;
            for (int64_t t : chain) if (e[t] > e[end] && e[t] < mn) mn = e[t];
            e[pit] = mn;
        }
        const double base = e[pit];
        const double drop = dtype_mode == 2 ? (double)((float)e[end] - (float)base) : e[end] - base;
        const int64_t L = (int64_t)chain.size();
        const double step = 1.0 / (double)(L - 1);                             // np.linspace(0, 1, L): arange * step, last = 1
        for (int64_t t = 0; t < L; t++) {
            const double f = (t == L - 1) ? 1.0 : (double)t * step;
            double v = base + f * drop;
            if (dtype_mode == 1) v = trunc(v);
            else if (dtype_mode == 2) v = (double)(float)v;
            e[chain[(size_t)t]] = v;
        }
    }
    *n_failed = failed;
    *iter_used = used;
    return 0;
}

}  // extern "C"
