"""The numpy / scipy statements the native conditioning loops (pydem_amd/csrc/cond_host.cpp) and the device
kernels (csrc/cond_device.hip, csrc/cond_paths.hip) were written from: region by region, pit by pit, one scipy.ndimage
call at a time like the reference (pydem/dem_processing.py:308-579, helpers pydem/utils.py:270-468).  TEST
INFRASTRUCTURE ONLY: the product never imports this module; tests compare the native host loops with it on random
tiles (tests/test_conditioning_golden.py, tools/soak_conditioning.py), and it is itself pinned bit for bit by the
g5_* / g7_* goldens captured from the reference."""
import warnings

import numpy as np
from scipy import ndimage

from pydem_amd.conditioning import _CROSS, _EIGHT, _RING, _SQRT2, _sea_mask


def fill_pit_artifacts(elev, maximum_pit_area=32.0, fill_flats_below_sea=False):
    """The same in numpy / scipy (what the native loop was written from; used by the tests and for float32)."""
    elev = np.asarray(elev)
    out = elev.copy()
    low = (ndimage.minimum_filter(elev, (3, 3)) >= elev) & _sea_mask(elev, fill_flats_below_sea)
    lab, _ = ndimage.label(low, structure=_EIGHT)
    nr, nc = elev.shape
    for k, box in enumerate(ndimage.find_objects(lab), 1):
        r0, r1, c0, c1 = box[0].start, box[0].stop, box[1].start, box[1].stop
        if r0 == 0 or c0 == 0 or r1 == nr or c1 == nc:
            continue                      # the one-pixel rim must lie inside the array (:414-415)
        win = (slice(r0 - 1, r1 + 1), slice(c0 - 1, c1 + 1))
        body = lab[win] == k
        size = int(body.sum())
        if size > maximum_pit_area:
            continue
        rim = ndimage.maximum_filter(body, (3, 3)) ^ body
        patch = elev[win]
        if np.all(patch[rim] - 1 == patch[body][0]):
            out[win] += (1 * body).astype(out.dtype)
    return out


def _neighbour_ring(mask):
    """Cells 8-adjacent to `mask` but not in it (utils.get_border_mask :342-370, including its shortcut:
    when the interior of the window is entirely region, everything else counts as border)."""
    inner = mask[1:-1, 1:-1]
    if inner.all() and inner.any():
        return ~mask
    grown = ndimage.binary_dilation(mask, structure=_EIGHT)
    return grown & ~mask


def _chamfer_distance(region, seeds):
    """Within-region (1, sqrt 2) chamfer distance from `seeds`, iterated exactly like utils.get_distance
    (:374-402): Jacobi sweeps over the whole window, stopping as soon as every region cell has *some*
    finite value -- not at convergence -- which is part of the reference's result."""
    big = float(region.size)
    d = np.full(region.shape, big)
    d[seeds] = 0
    for _ in range(region.size):
        straight = ndimage.minimum_filter(d, footprint=_CROSS) + 1
        diagonal = ndimage.minimum_filter(d, (3, 3)) + _SQRT2
        best = np.minimum(straight[region], diagonal[region])
        d[region] = np.minimum(best, d[region])
        if (d[region] < big).all():
            break
    return d


def _centre_cell(region):
    """Region cell nearest to the centre of mass (utils.find_centroid :450-468)."""
    cy, cx = ndimage.center_of_mass(region)
    cells = np.argwhere(region)
    i, j = cells[np.argmin(np.linalg.norm(cells - (cy, cx), axis=1))]
    return i, j


def _fill_one_flat(roi, out, region, edge, source_tol, peaks, pits):
    """One labelled flat inside its one-pixel-grown window (reference _fill_flat :308-394).  `roi` is the
    unmodified surface, `out` the window of the surface being built.  The reference's recursive pass over
    flats created by the interpolation writes into a scratch copy that is then dropped (:389-394), so it
    has no effect on the result and is not repeated here."""
    level = roi[region][0]
    if roi.size <= 9 and region.sum() == 1:                     # single pixel in a tiny window (:312-325)
        higher = roi > level
        n_high = int(higher.sum())
        if n_high == roi.size - 1:
            return                                              # a true pit: leave it
        if n_high > 0:
            out[region] += min(1.0, roi[higher].min() - level) - 0.01
        elif peaks:
            out[region] += 0.5
        return
    ring = _neighbour_ring(region)
    drain = ring & (roi == level)
    source = ring & (roi > level)
    pinned = None                                               # cells whose value is set, not interpolated
    if source.any():                                            # gentle uphill rim (:343-347)
        lowest = roi[source].min()
        top = min(level + 1.0, lowest)
        source &= (roi <= lowest + source_tol)
    elif peaks:                                                 # summit plateau: drain away from its centre (:348-354)
        top = level + 0.5
        ci = _centre_cell(region)
        out[ci] = top
        source[ci] = True
        pinned = source
    else:
        return
    if drain.any():
        pass
    elif (region & edge).any():                                 # river bed leaving through the tile edge (:362-366)
        pinned = drain = region & edge
        if not (region & ~drain).any():
            return
    elif pits:                                                  # closed depression: drain towards its centre (:367-371)
        ci = _centre_cell(region)
        drain[ci] = True
        pinned = drain
    else:
        return
    d_high = _chamfer_distance(region, source)
    d_low = _chamfer_distance(region, drain)
    target = region if pinned is None else region & ~pinned
    out[target] = (top * d_low[target] ** 2 + level * d_high[target] ** 2) / (d_low[target] ** 2 + d_high[target] ** 2)


def fill_flats(elev, maximum_pit_area=32.0, fill_flats_below_sea=False, fill_flats_source_tol=1,
                     fill_flats_peaks=True, fill_flats_pits=True):
    """The same in numpy / scipy (what the native loop was written from; used by the tests)."""
    if maximum_pit_area:
        elev = fill_pit_artifacts(elev, maximum_pit_area, fill_flats_below_sea)
    data = np.ma.filled(np.asarray(elev).astype('float64'), np.nan)
    built = data.copy()
    edge = np.ones(data.shape, bool)
    edge[1:-1, 1:-1] = False
    flat = (ndimage.minimum_filter(data, (3, 3)) >= data) & _sea_mask(data, fill_flats_below_sea)
    flat[0, 0] = flat[-1, 0] = flat[0, -1] = flat[-1, -1] = False          # corners never (:569-572)
    lab, _ = ndimage.label(flat, structure=_EIGHT)
    nr, nc = data.shape
    for k, box in enumerate(ndimage.find_objects(lab), 1):
        win = (slice(max(0, box[0].start - 1), min(nr, box[0].stop + 1)),
               slice(max(0, box[1].start - 1), min(nc, box[1].stop + 1)))
        _fill_one_flat(data[win], built[win], lab[win] == k, edge[win], fill_flats_source_tol,
                       fill_flats_peaks, fill_flats_pits)
    return built


def _ring_of(cells, in_set, nr, nc):
    """8-neighbours of `cells` (flat ids) that are not members of the set."""
    out = set()
    for c in cells:
        i, j = divmod(c, nc)
        for di in (-1, 0, 1):
            ii = i + di
            if ii < 0 or ii >= nr:
                continue
            for dj in (-1, 0, 1):
                jj = j + dj
                if (di or dj) and 0 <= jj < nc:
                    t = ii * nc + jj
                    if t not in in_set:
                        out.add(t)
    return out


def _mean_dx(dX, i1, i2):
    if i1 == i2:                                                 # _get_dX_mean :1993-1997
        return dX[min(i1, dX.size - 1)]
    return dX[min(i1, i2):max(i1, i2)].mean()


def pit_drain_paths(elev, dX, dY, drain_pits_max_iter=300, drain_pits_max_dist=32, drain_pits_max_dist_XY=None,
                          fill_flats_below_sea=False):
    """The same in numpy (what the native loop was written from; used by the tests and for non-float64 surfaces)."""
    nr, nc = elev.shape
    e = elev.ravel()                                             # view: edits land in `elev`
    lows = (ndimage.minimum_filter(elev, footprint=_RING).ravel() > e) & _sea_mask(e, fill_flats_below_sea)
    pit_ids = np.where(lows)[0]
    order = np.argsort(e[pit_ids])                               # same call as the reference (:450): same tie order
    failed = 0
    used = 0
    for pit in pit_ids[order]:
        pit = int(pit)
        area = {pit}
        trail = [pit]
        floor = e[pit]
        outlet = None
        rim = _ring_of([pit], area, nr, nc)
        it = 0
        for it in range(drain_pits_max_iter):
            if not rim:
                break
            rim_ids = np.fromiter(rim, dtype='int64', count=len(rim))
            rim_ids.sort()
            heights = e[rim_ids]
            lowest = heights.min()
            at_lowest = rim_ids[heights == lowest]
            if lowest < floor:
                outlet = at_lowest
                break
            fresh = at_lowest.tolist()
            trail += fresh
            area.update(fresh)
            rim.difference_update(fresh)
            rim |= _ring_of(fresh, area, nr, nc)
        if outlet is None:
            failed += 1
            continue
        used = max(used, it + 1)
        ip, jp = divmod(pit, nc)
        oi, oj = np.divmod(outlet, nc)
        if drain_pits_max_dist:                                  # index-space reach (:485-493)
            near = np.sqrt((ip - oi) ** 2 + (jp - oj) ** 2) <= drain_pits_max_dist
            if not near.any():
                failed += 1
                continue
            outlet, oi, oj = outlet[near], oi[near], oj[near]
        run = np.array([_mean_dx(dX, ip, int(a)) * (jp - int(b)) for a, b in zip(oi, oj)])
        rise = np.array([dY[min(ip, int(a)):max(ip, int(a))].sum() for a in oi])
        reach = np.sqrt(run ** 2 + rise ** 2)
        if drain_pits_max_dist_XY:                               # metric reach (:502-508)
            near = reach <= drain_pits_max_dist_XY
            if not near.any():
                failed += 1
                continue
            outlet, reach = outlet[near], reach[near]
        if outlet.size > 1:
            outlet = outlet[reach == reach.min()]
        end = int(outlet[0])
        # prune the trail, walking back from the outlet, to an 8-connected chain (:516-532)
        chain = trail + [end]
        ci, cj = [list(v) for v in np.unravel_index(chain, (nr, nc))]
        k = len(chain) - 2
        while k > 0:
            if abs(ci[k] - ci[k + 1]) <= 1 and abs(cj[k] - cj[k + 1]) <= 1:
                k -= 1
            else:
                del chain[k], ci[k], cj[k]
                k = min(k, len(chain) - 2)
            if chain[k] == pit:
                break
        # elevations fall linearly along the chain (:535-539)
        if e[pit] < e[end]:
            along = e[chain]
            e[pit] = along[along > e[end]].min()
        drop = e[end] - e[pit]
        e[chain] = e[pit] + np.linspace(0, 1, len(chain)) * drop
    if failed:
        warnings.warn("Warning %d pits had no place to drain to in this chunk" % failed)
    return elev, failed, used
