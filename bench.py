#!/usr/bin/env python
"""bench.py -- headline benchmark: Mcells/s of the per-tile terrain hot path
(slope + aspect + flats + section/proportion + UCA sweep + TWI) on synthetic DEM tiles.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W        (any launcher that sets RANK / LOCAL_RANK / WORLD_SIZE)

Workload (BASELINE.json: the metric is quoted on the 16384x16384 fp64 tile, configs[2]): every
rank owns ONE 16384x16384 float64 fractal tile (deterministic generator, evaluated at the tile's
global mosaic coordinates, generated on the device so the input is HBM-resident before the timed
region).  One step = one full pass of the hot path over the resident tile
(pydem_slopes_directions + pydem_uca + pydem_twi through the C-ABI).  Tiles are independent
units: weak scaling, no data-path collective in the timed region (the cross-tile edge fix-up is
a separate, latency-bound exchange -- DESIGN.md).

Prints ONE JSON line on rank 0.  `roofline` describes the DOMINANT stage of the step, the UCA sweep (the
tile-pass kernels k_sweep_tiles*): algorithmic bytes 40 B per cell (12 read: graph word 4 + proportion 8;
28 written: area 8, the two contributions 16, level stamp 4 -- DESIGN.md section 4) over the stage time that
the C-ABI brackets with hipEvents on the tile's own HIP stream in every timed step.  `roofline_stages` lists
the same three numbers (algorithmic bytes, time, PMC traffic from the committed rocprofv3 passes) for the
stencil, the pit search, the sweep and TWI; `roofline_stencil` is the slope/aspect kernel alone (24 B/cell:
read elev 8 + write mag 8 + direction 8, SURVEY.md section 8d -- the kernel BASELINE.json's 40 % target is
about); its `back_to_back_ms` re-launches the kernel `--roof-iters` times in a row after the run.  `cpu_baseline` times the CPU oracle (a port of the reference algorithm,
bit-exact against golden vectors of the reference) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC (RCCL across processes)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
STENCIL_BYTES_PER_CELL = 24.0
E2E_BYTES_PER_CELL = 40.0      # read elev 8 + write mag, direction, uca, twi 32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--size', type=int, default=16384, help='tile edge (cells)')
    ap.add_argument('--drain-pits', type=int, default=int(os.environ.get('PYDEM_BENCH_DRAIN_PITS', '1')))
    ap.add_argument('--roof-iters', type=int, default=20)
    ap.add_argument('--cpu-sample', type=int, default=4096, help='edge of the CPU-baseline sample tile (0 = skip)')
    ap.add_argument('--host-to-host', type=int, default=1, help='1: also time the drop-in call from a host array to the TWI on the host (N = 1; reported beside `value`)')
    ap.add_argument('--config', type=int, default=3, choices=[2, 3, 5],
                    help='BASELINE.json config: 3 = the headline line (default); 2 = 4096^2 stencil kernel only; 5 = 8192^2 int16 with the '
                         'reference defaults (conditioning on the device).  2 and 5 are single-GPU side lines, stored under profiles/')
    return ap.parse_args()


def make_options(_ffi, drain_pits):
    o = _ffi.Options()
    o.drain_pits = int(drain_pits)
    o.drain_pits_min_border = 0
    o.drain_pits_max_iter = 300
    o.drain_pits_max_dist = 32
    o.drain_pits_max_dist_XY = float('nan')
    o.apply_uca_limit_edges = 0
    o.apply_twi_limits = 0
    o.apply_twi_limits_on_uca = 0
    o.circular_ref_maxcount = 50
    o.uca_saturation_limit = 32.0
    o.twi_min_slope = 1e-3
    o.twi_min_area = float('inf')
    return o


def reference_calibration():
    """port / reference speed ratio on the bench workload, measured in the build container against the unmodified
    reference (oracle/ref_harness/calibrate_cpu_baseline.py -> profiles/r02_cpu_calibration.json); None if absent."""
    fn = os.path.join(ROOT, 'profiles', 'r02_cpu_calibration.json')
    if not os.path.exists(fn):
        return None
    with open(fn) as f:
        return json.load(f)


def cpu_baseline(size, seed, drain_pits):
    """The oracle (port of the reference's algorithm) on one size x size tile of the same generator."""
    from oracle import oracle as O
    z = O.synth_fractal(size, size, seed=seed)
    t0 = time.perf_counter()
    o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=bool(drain_pits))
    o.calc_twi()
    dt = time.perf_counter() - t0
    out = {"value": size * size / dt / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
           "sample": "one %dx%d fp64 fractal tile (seed %d), full path, %.1f s, single thread "
                     "(the reference is single-threaded per tile); host has %d cores"
                     % (size, size, seed, dt, os.cpu_count())}
    cal = reference_calibration()
    if cal is not None and drain_pits:
        # what the unmodified reference would do on this host: the port's speed divided by the measured ratio
        r = cal['port_over_reference_bench']
        out["reference_equiv"] = {"value": out["value"] / r, "unit": "Mcells/s", "port_over_reference": r,
                                  "range": cal['port_over_reference_range_pits'],
                                  "source": "profiles/r02_cpu_calibration.json (build container, unmodified reference vs port, same workload)"}
    return out


def run_config2(args):
    """BASELINE.json config 2: 4096 x 4096 fp64 fractal, the slope / aspect stencil kernel only."""
    from pydem_amd import DEMProcessor
    n = 4096
    dp = DEMProcessor.from_synthetic((n, n), dict(seed=0), dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False)
    dp.run_slopes_directions()
    tile = dp._tile
    tile.bench_stencil(max(args.warmup, 20))
    iters = max(args.steps, 200)
    ms = tile.bench_stencil(iters)
    cells = float(n) * n
    achieved = STENCIL_BYTES_PER_CELL * cells / (ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "Mcells/s (slope+aspect 3x3 stencil kernel only); % HBM roofline", "value": cells / (ms * 1e-3) / 1e6, "unit": "Mcells/s",
        "n_gpus": 1, "steps": iters, "warmup": max(args.warmup, 20), "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 2: 4096x4096 fp64 fractal tile (seed 0, dX=dY=30 m), k_stencil_march only, %d back-to-back "
                               "launches after %d warm-up" % (iters, max(args.warmup, 20)), "tile": [n, n]},
        "roofline": {"bound": "hbm", "kernel": "k_stencil_march", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": STENCIL_BYTES_PER_CELL * cells,
                     "avg_kernel_ms": ms, "algorithmic_bytes_per_cell": STENCIL_BYTES_PER_CELL}}))


def run_config5(args):
    """BASELINE.json config 5: 8192 x 8192 int16 SRTM-like tile, reference defaults (fill_flats, drain_pits_path, drain_pits).
    A step = conditioning (artefacts, flats, pit drain paths) + slopes / directions + UCA + TWI on the device; the raw int16
    tile is uploaded again at the start of every step (the conditioning rewrites the resident surface) and that copy is
    timed separately and NOT part of `value`."""
    import numpy as np
    from pydem_amd import DEMProcessor, synth
    n = args.size if args.size != 16384 else 8192
    z = synth.srtm_int16(n, n, seed=3)
    dp = DEMProcessor(elev=z, dX=30.0, dY=30.0)
    stages = {}

    def step():
        t0 = time.perf_counter()
        dp.elev = z
        dp._ensure_tile(); dp._push('elev'); dp._tile.synchronize()
        t1 = time.perf_counter()
        dp.fill_flats = True; dp.drain_pits_path = True
        dp.calc_fill_flats()
        dp._tile.synchronize()
        t2 = time.perf_counter()
        on_dev = dp._pit_paths_on_device()
        if on_dev is None:
            raise SystemExit("bench --config 5: the pit drain paths fell back to the host loop")
        dp._tile.synchronize()
        t3 = time.perf_counter()
        dp.fill_flats = False; dp.drain_pits_path = False
        dp.run_slopes_directions(); dp.run_uca(); dp.run_twi()
        dp._tile.synchronize()
        t4 = time.perf_counter()
        stages.update(h2d_ms=(t1 - t0) * 1e3, fill_flats_ms=(t2 - t1) * 1e3, pit_paths_ms=(t3 - t2) * 1e3, terrain_ms=(t4 - t3) * 1e3,
                      pit_path_rounds=on_dev[2], pits_without_outlet=on_dev[0])
        return t4 - t1
    for _ in range(max(1, args.warmup)):
        step()
    dts = [step() for _ in range(args.steps)]
    dt = sum(dts) / len(dts)
    tm = dp._tile.timings()
    cells = float(n) * n
    st_ms = tm['stencil_kernel_ms']
    achieved = STENCIL_BYTES_PER_CELL * cells / (st_ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "Mcells/s (conditioning+slope+aspect+UCA+TWI end-to-end) per tile; % HBM roofline", "value": cells / dt / 1e6, "unit": "Mcells/s",
        "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE config 5: %dx%d int16 SRTM-like tile (seed 3, lakes flooded to exact plateaus), reference defaults: "
                               "fill_flats + drain_pits_path + drain_pits; conditioning, slopes_directions, uca, twi on the device" % (n, n),
                   "tile": [n, n]},
        "roofline": {"bound": "hbm", "kernel": "k_stencil_march", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": STENCIL_BYTES_PER_CELL * cells,
                     "avg_kernel_ms": st_ms, "algorithmic_bytes_per_cell": STENCIL_BYTES_PER_CELL},
        "stages_ms": dict({k: tm[k] for k in ('stencil_kernel_ms', 'flats_ms', 'graph_ms', 'pits_ms', 'sweep_ms', 'twi_ms')}, **stages),
        "not_in_value": {"h2d_ms": stages['h2d_ms'], "what": "upload of the raw int16 tile (pageable host memory) at the start of the step"}}))


def host_to_host(dp_resident, drain_pits):
    """The drop-in call as a user makes it -- DEMProcessor(elev=<host array>).calc_twi() -> TWI on the host -- timed once on
    the bench tile (outside the timed region; PCIe-inclusive, reported beside `value`, never as it): upload of the
    elevations, the device stages, download of the TWI through pageable host memory."""
    import gc
    import numpy as np
    from pydem_amd import DEMProcessor
    z = np.array(dp_resident.elev)                       # the tile as a host array (the bench generated it on the device)
    best = None
    for rep in range(3):                                 # (the first call of a process pays one-off allocations)
        t0 = time.perf_counter()
        dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=drain_pits)
        dp._ensure_tile(); dp._push('elev'); dp._tile.synchronize()
        t1 = time.perf_counter()
        dp.run_twi(); dp._tile.synchronize()
        t2 = time.perf_counter()
        twi = dp.calc_twi()                              # (the stages are done: this is the download)
        t3 = time.perf_counter()
        tm = dp.timings
        dev = sum(tm[k] for k in ('slopes_directions_ms', 'flats_ms', 'graph_ms', 'pits_ms', 'sweep_ms', 'twi_ms'))
        rec = {"ms": (t3 - t0) * 1e3, "upload_ms": (t1 - t0) * 1e3, "stages_wall_ms": (t2 - t1) * 1e3, "device_stages_ms": dev,
               "twi_download_ms": (t3 - t2) * 1e3, "Mcells_per_s": z.size / (t3 - t0) / 1e6,
               "what": "DEMProcessor(elev=host float64 array).calc_twi() -> host array, pageable memory; best of 3 calls"}
        if os.environ.get('PYDEM_BENCH_DEBUG'):
            print('host_to_host rep %d: %r' % (rep, rec), file=sys.stderr, flush=True)
        if best is None or rec["ms"] < best["ms"]:
            best = rec
        del dp, twi
        gc.collect()                                     # (the tile of this call is released here, not inside the next call)
    return best


def csrc_hashes():
    """sha256 per kernel source file (pydem_amd/csrc/*): the identity of the kernels a PMC profile was collected from"""
    import hashlib
    d = os.path.join(ROOT, 'pydem_amd', 'csrc')
    return {f: hashlib.sha256(open(os.path.join(d, f), 'rb').read()).hexdigest() for f in sorted(os.listdir(d))}


def pmc_traffic(kernels, size, sources=()):
    """HBM bytes per STEP of the kernels whose names contain one of `kernels` (a name or a tuple of names), from the
    committed rocprofv3 PMC passes (profiles/r*_pmc_fetch_write_16384*.csv: separate FETCH_SIZE / WRITE_SIZE runs of
    this same command with --steps 1 --warmup 0, so `dispatches` is per step).  FETCH_SIZE counts half of the bytes of
    wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section; calibrated on k_twi in profiles/README.md), so
    bytes = (2 * FETCH_KiB + WRITE_KiB) * 1024, summed over the dispatches.  None when no profile of this tile size is
    present (PMC counters cannot be read from inside the timed process) -- or when the profile is STALE: every profile
    carries the sha256 of the kernel sources it was collected from (<csv>.meta.json, tools/csrc_stamp.py); `sources` are
    the files of this stage's kernels, and a profile whose hashes of them (or of internal.h) differ from the tree's is not
    quoted."""
    import csv
    import glob
    if size != 16384:
        return None
    if isinstance(kernels, str):
        kernels = (kernels,)
    best = None
    now = csrc_hashes()
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_fetch_write_16384*.csv'))):
        meta = fn[:-4] + '.meta.json'
        if not os.path.exists(meta):
            continue
        then = json.load(open(meta)).get('csrc_sha256', {})
        if any(then.get(f) != now.get(f) for f in tuple(sources) + ('internal.h',)):
            continue
        tot = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0}
        seen = set()
        for row in csv.DictReader(open(fn)):
            if any(k in row['kernel'] for k in kernels) and row['counter'] in tot:
                # the stencil is re-launched for `back_to_back_ms` after the step: one dispatch per step counts
                disp = 1 if 'k_stencil' in row['kernel'] else int(row['dispatches'])
                tot[row['counter']] += float(row['mean_per_dispatch_KB']) * disp
                seen.add(row['counter'])
        if len(seen) == 2:
            best = (2.0 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024.0
    return best


def stencil_valu_insts(size):
    """Vector instructions of ONE launch of the stencil kernel (SQ_INSTS_VALU summed over the chip) from the committed SQ pass
    (profiles/r*_pmc_sq_stencil_16384.csv, tools/collect_profiles.sh), only from a profile stamped with the current stencil.hip
    (<csv>.meta.json); None otherwise."""
    import csv
    import glob
    if size != 16384:
        return None
    best = None
    now = csrc_hashes()
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_sq_stencil_16384.csv'))):
        meta = fn[:-4] + '.meta.json'
        if not os.path.exists(meta):
            continue
        then = json.load(open(meta)).get('csrc_sha256', {})
        if any(then.get(f) != now.get(f) for f in ('stencil.hip', 'internal.h')):
            continue
        got = {}
        for row in csv.DictReader(open(fn)):
            if 'k_stencil_march' in row['kernel'] and row['counter'] in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU') and int(row['dispatches']) > 1:
                got[row['counter']] = float(row['mean_per_dispatch_KB'])           # (the column holds plain counts for SQ counters)
        if 'SQ_INSTS_VALU' in got:
            best = got
    return best


def stencil_issue_model():
    """Issue cycles per vector instruction of the marching loop from the committed instruction histogram
    (profiles/r*_stencil_isa_histogram.txt, tools/stencil_isa_histogram.sh: 64-bit / fp64 instructions 4 cycles, 32-bit ones 2,
    fp64 transcendentals 8 -- MI355X_MICROARCH.md, SIMD-32), only from a histogram of the current stencil.hip; None otherwise."""
    import glob
    import hashlib
    now = hashlib.sha256(open(os.path.join(ROOT, 'pydem_amd', 'csrc', 'stencil.hip'), 'rb').read()).hexdigest()
    best = None
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_stencil_isa_histogram.txt'))):
        for line in open(fn):
            if line.startswith('#json '):
                d = json.loads(line[6:])
                if d.get('stencil_sha256') == now:
                    best = d
    return best


N_CU, N_SIMD, VALU_CYCLES_PER_INST, PEAK_CLOCK_HZ = 256, 256 * 4, 4.0, 2.4e9      # MI355X: 256 CUs x 4 SIMDs; a wave64 fp64 instruction issues over 4 cycles (32-bit ones over 2: stencil_issue_model)


# stages of the step for `roofline_stages`: (name, kernels of the stage, algorithmic bytes per cell, timing key, what the bytes are,
# source files of the kernels: `traffic` is only quoted from a PMC profile of these very sources)
STAGES = [
    ('stencil', ('k_stencil_march', 'k_stencil_perimeter'), 24.0, 'stencil_kernel_ms', 'read elev 8 + write mag 8 + direction 8',
     ('stencil.hip',)),
    ('pits', ('k_pits_', 'k_pitmask', 'k_pit_keys', 'k_pit_gather', 'radix_sort'), 9.4, 'pits_ms',
     'read elev 8 + flats 1 per cell + write 16 B per pit edge (6.49 M edges on the bench tile: 0.4 B/cell)', ('pits.hip',)),
    ('sweep', ('k_sweep_tiles', 'k_sweep_sym', 'k_sym_candidates', 'k_sym_finish', 'k_pit_stash', 'k_uca_finalize'), 40.0, 'sweep_ms',
     'read graph word 4 + proportion 8, write area 8 + two contributions 16 + level stamp 4 per cell', ('uca.hip', 'uca_sym.inl')),
    ('twi', ('k_twi',), 24.0, 'twi_ms', 'read uca 8 + mag 8, write twi 8', ('uca.hip',)),
]


def tile_specs(world, n, m, px=30.0):
    """One tile per rank, laid out as a (<=2) x (<=4) mosaic with a one-pixel overlap (BASELINE.json
    config 4 layout); elevations come from the device generator at global mosaic coordinates."""
    ncol = min(world, 4)
    specs = []
    for t in range(world):
        r, c = t // ncol, t % ncol
        row0, col0 = r * (n - 1), c * (m - 1)
        specs.append({'shape': (n, m), 'synth': dict(seed=1, row0=row0, col0=col0),
                      'bounds': (col0 * px, -(row0 + n) * px, (col0 + m) * px, -row0 * px)})
    return specs


def main():
    args = parse()
    if args.config != 3:
        import warnings
        warnings.simplefilter('ignore')
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            raise SystemExit("bench --config %d is a single-GPU line" % args.config)
        return run_config2(args) if args.config == 2 else run_config5(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: one process per GPU, started here (rank 0 prints the line)
        from pydem_amd import rendezvous
        sys.exit(rendezvous.spawn_ranks([sys.executable] + sys.argv, args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import warnings
    warnings.simplefilter('ignore')
    from pydem_amd import _ffi, process_manager
    n = m = args.size
    ndev = _ffi.device_count()
    device = local_rank % ndev
    pm = process_manager.ProcessManager(elev_source_files=tile_specs(world, n, m), elev_conditioned=True,
                                        dem_proc_kwargs={'drain_pits': bool(args.drain_pits)}, devices=[device],
                                        keep_first_pass_uca=False, n_workers=world,
                                        edge_mode=('pool' if world > 1 else 'reference'))
    exchange = "in-process"
    group = None
    rccl_ranks = None
    if world > 1:
        shared = os.environ.get('PYDEM_BENCH_SHARED_GPU') == '1'
        if ndev < world and not shared:
            # RCCL refuses two ranks on one device, and a communicator that half of the ranks could not join hangs the others:
            # say so before anything is started
            raise SystemExit("bench: --gpus %d needs %d GPUs, this box has %d (one rank per GPU over RCCL).  PYDEM_BENCH_SHARED_GPU=1 runs "
                             "the ranks on the GPUs there are with the strips summed over sockets -- a functional check of the "
                             "multi-rank path, not a scaling number." % (world, world, ndev))
        # process-group plumbing only (hands the RCCL id around: pydem_amd/rendezvous.py); the strips travel over RCCL
        from pydem_amd import parallel, rendezvous
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        group = rendezvous.SocketGroup(rank, world)
        rccl = None
        if not shared:
            try:
                rccl = parallel.make_rccl_transport(pm, device, group)
            except Exception as e:      # keep the scaling run alive and say so in the JSON line
                sys.stderr.write("bench: rank %d: RCCL transport unavailable (%s)\n" % (rank, e))
        # every rank must take the same path: one failed communicator sends all of them to the host fallback
        if all(group.all_gather_object(rccl is not None)):
            pm.transport = rccl
            exchange = "rccl"
            rccl_ranks = rccl.comm.count()
            if rccl_ranks != world:
                raise SystemExit("bench: rank %d: the RCCL communicator has %d ranks, WORLD_SIZE is %d" % (rank, rccl_ranks, world))
        else:
            pm.transport = parallel.DistTransport(pm, group)
            exchange = "socket-host-fallback"
    pm.compute_grid()
    pm.process_elevation()          # tiles are generated on their GPU: HBM-resident before the timed region
    mine = [i for i in range(pm.n_inputs) if pm.transport.owns(i)]
    phase = {}

    def step():
        t0 = time.perf_counter()
        pm.process_aspect_slope()
        pm.process_uca()
        for i in mine:
            pm.tiles[i]._tile.synchronize()
        t1 = time.perf_counter()
        pm.process_uca_edges()
        t2 = time.perf_counter()
        for i in mine:
            pm.tiles[i].find_flats()
            pm.tiles[i].run_twi()
        t3 = time.perf_counter()
        phase['tile_ms'] = (t1 - t0 + t3 - t2) * 1e3
        phase['edge_fixup_ms'] = (t2 - t1) * 1e3
        tmk = pm.tiles[mine[0]]._tile.timings()
        stencil_ms.append(tmk['stencil_kernel_ms'])
        for _, _, _, key, _, _ in STAGES:
            stage_ms.setdefault(key, []).append(tmk[key])

    def barrier():
        for i in mine:
            pm.tiles[i]._tile.synchronize()
        if world > 1:
            pm.transport.barrier()

    stencil_ms, stage_ms = [], {}
    pm.edge_board_digest = world > 1          # (warm-up only: a download of the replicated edge board per fix-up)
    for _ in range(args.warmup):
        step()
    barrier()
    pm.edge_board_digest = False
    per_rank = None
    if world > 1:
        # Every rank runs the SAME schedule from replicated numbers (the wave selection is a kernel on every rank): before the
        # timed region the ranks compare what the warm-up left behind -- waves / rounds / tie-breaks and the replicated board --
        # and stop with a message instead of drifting apart inside a collective later.
        mine_state = (getattr(pm, 'edge_schedule_digest', None), getattr(pm, 'edge_board_sha256', None) if args.warmup > 0 else None)
        states = group.all_gather_object(mine_state)
        if any(st != states[0] for st in states):
            raise SystemExit("bench: the ranks disagree after the warm-up: (waves, rounds, tie-breaks), sha256 of the edge board per rank = %r" % (states,))
    del stencil_ms[:]
    stage_ms.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = pm.transport.allreduce_max(dt)
        per_rank = group.all_gather_object({"rank": rank, "edge_waves": int(pm.edge_waves), "edge_rounds_run_here": len(pm.edge_round_log),
                                            "edge_host_looks": int(getattr(pm, 'edge_host_looks', 0)),
                                            "edge_queued_batches": int(getattr(pm, 'edge_queued_batches', 0)),
                                            "edge_fixup_ms": phase.get('edge_fixup_ms'), "tile_ms": phase.get('tile_ms')})
    tile = pm.tiles[mine[0]]._tile
    tm = tile.timings()
    if tm['n_unresolved']:
        raise SystemExit("bench: %d cells unresolved (cyclic drainage) -- result invalid" % tm['n_unresolved'])

    if rank == 0:
        cells = float(n) * m
        value = world * cells * args.steps / dt / 1e6
        st_b2b = tile.bench_stencil(args.roof_iters) if args.roof_iters > 0 else None
        st_ms = sum(stencil_ms) / len(stencil_ms)
        achieved = STENCIL_BYTES_PER_CELL * cells / (st_ms * 1e-3) / 1e9
        ms_step = dt / args.steps * 1e3
        stages = []
        for name, kernels, bpc, key, what, srcs in STAGES:
            ms = sum(stage_ms[key]) / len(stage_ms[key])
            if ms <= 0:
                continue
            gbs = bpc * cells / (ms * 1e-3) / 1e9
            stages.append({"stage": name, "kernels": list(kernels), "ms": ms, "share_of_step": ms / ms_step,
                           "algorithmic_bytes": bpc * cells, "algorithmic_bytes_per_cell": bpc, "bytes_are": what,
                           "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "traffic": pmc_traffic(kernels, n, srcs)})
        dom = max(stages, key=lambda d: d["ms"])
        sq = stencil_valu_insts(n) or {}
        valu, salu = sq.get('SQ_INSTS_VALU'), sq.get('SQ_INSTS_SALU')
        model = stencil_issue_model()
        cpi = model['valu_cycles_per_inst'] if model else VALU_CYCLES_PER_INST      # (no histogram of this stencil.hip: every instruction charged as fp64)
        valu_floor_ms = valu * cpi / N_SIMD / PEAK_CLOCK_HZ * 1e3 if valu else None
        salu_floor_ms = salu / N_CU / PEAK_CLOCK_HZ * 1e3 if salu else None        # one scalar instruction per CU and cycle
        out = {
            "metric": "Mcells/s (slope+aspect+UCA+TWI end-to-end) per tile; % HBM roofline",
            "value": value, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%dx%d fp64 fractal tile per GPU (seed 1, dX=dY=30 m, %d-tile mosaic with 1-pixel "
                                   "overlap), fill_flats=False, drain_pits_path=False, drain_pits=%s: "
                                   "slopes_directions + uca + cross-tile edge fix-up + twi"
                                   % (n, m, world, bool(args.drain_pits)),
                       "tile": [n, m], "tiles_per_gpu": 1, "parallelism": "tile-per-gpu x%d" % world,
                       "edge_exchange": exchange},
            # the dominant stage of the step (largest share of ms_per_step); stage time = hipEvents on the tile's stream
            "roofline": {"bound": "hbm", "kernel": "/".join(dom["kernels"]) + " (stage '%s')" % dom["stage"], "achieved": dom["achieved"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": dom["traffic"],
                         "algorithmic_bytes": dom["algorithmic_bytes"], "avg_kernel_ms": dom["ms"],
                         "algorithmic_bytes_per_cell": dom["algorithmic_bytes_per_cell"], "share_of_step": dom["share_of_step"],
                         "note": "stage = the kernels listed, bracketed by hipEvents on the tile's stream in every timed step; "
                                 "latency-bound tile passes (DESIGN.md section 4), not a streaming kernel"},
            "roofline_stages": stages,
            # the slope / aspect kernel alone: the kernel BASELINE.json's 40 % target is about
            "roofline_stencil": {"bound": "hbm", "kernel": "k_stencil_march", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_stencil_march", n, ('stencil.hip',)),
                                 "algorithmic_bytes": STENCIL_BYTES_PER_CELL * cells,
                                 "avg_kernel_ms": st_ms, "back_to_back_ms": st_b2b,
                                 "algorithmic_bytes_per_cell": STENCIL_BYTES_PER_CELL,
                                 # the kernel is bound by its vector instructions (fp64, no packed forms): the same launch against the
                                 # issue rate of the vector ALUs = instructions x 4 cycles / 1024 SIMDs / 2.4 GHz / measured time
                                 # (since round 6 with the cycles of the instruction mix -- 32-bit vector instructions issue over 2 cycles, fp64 over 4 --
                                 # and with the SCALAR unit beside it: the mask algebra runs there, the kernel is bound by both issue ports)
                                 "valu_insts": valu, "valu_cycles_per_inst": cpi if valu else None, "valu_floor_ms": valu_floor_ms,
                                 "valu_roofline_frac": (valu_floor_ms / st_ms) if valu else None,
                                 "valu_roofline_frac_back_to_back": (valu_floor_ms / st_b2b) if (valu and st_b2b) else None,
                                 "salu_insts": salu, "salu_floor_ms": salu_floor_ms,
                                 "salu_roofline_frac": (salu_floor_ms / st_ms) if salu else None,
                                 "salu_roofline_frac_back_to_back": (salu_floor_ms / st_b2b) if (salu and st_b2b) else None},
            "end_to_end_GBs": E2E_BYTES_PER_CELL * world * cells * args.steps / dt / 1e9,
            "stages_ms": dict({k: tm[k] for k in ('slopes_directions_ms', 'stencil_kernel_ms', 'flats_ms', 'graph_ms',
                                                   'pits_ms', 'sweep_ms', 'twi_ms')}, **phase),
            "sweep": {"rounds": tm['sweep_rounds'], "kernel_launches": tm['sweep_kernel_launches'],
                      "n_flats": tm['n_flats'], "n_pit_edges": tm['n_pit_edges'], "edge_rounds": pm.edge_rounds,
                      "edge_waves": pm.edge_waves, "edge_mode": 'pool' if world > 1 else 'reference'},
            "device_bytes": tile.device_bytes(),
        }
        if world > 1:
            out["rccl_ranks"] = rccl_ranks          # ncclCommCount of the communicator the strips travelled over (None: socket fallback)
            out["per_rank"] = per_rank
            out["sweep"]["edge_wave_graphs"] = bool(getattr(pm, 'edge_wave_graphs', False))
        if args.cpu_sample:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, 1, args.drain_pits)
        if world == 1 and args.host_to_host:
            out["host_to_host"] = host_to_host(pm.tiles[mine[0]], bool(args.drain_pits))
        print(json.dumps(out))
    if world > 1:
        pm.transport.barrier()
        group.barrier()
        group.close()


if __name__ == '__main__':
    main()
