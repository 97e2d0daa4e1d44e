/*
 * pydem_hip.h -- C-ABI of libpydem_hip.so: the MI355X (gfx950) implementation of pyDEM's
 * per-tile terrain hot path.  Plain pointers and sizes only; every call returns 0 on success
 * and a negative code on failure, with the message available from pydem_hip_last_error().
 *
 * What each entry point replaces in the reference (creare-com/pydem v1.2.1; paths relative to
 * the reference checkout):
 *
 *   pydem_slopes_directions   DEMProcessor.calc_slopes_directions  pydem/dem_processing.py:587-619
 *                             (= _tarboton_slopes_directions :1753-1903, _find_flats_edges :657-680)
 *   pydem_find_flats          DEMProcessor.find_flats              pydem/dem_processing.py:305-306
 *   pydem_uca                 DEMProcessor.calc_uca (uca_init=None) pydem/dem_processing.py:682-776,
 *                             _calc_uca_chunk :864-987, _calc_uca_section_proportion :1021-1070,
 *                             _mk_adjacency_matrix :1072-1153, _mk_connectivity_pits :1269-1382 and
 *                             the native loop cyutils.drain_area   pydem/cyfuncs/cyutils.pyx:78-187
 *   pydem_uca_edge_update     DEMProcessor.calc_uca(uca_init=, edge_init_data=) :724-771,
 *                             _calc_uca_chunk_update :778-862, cyutils.drain_connections
 *                             pydem/cyfuncs/cyutils.pyx:35-72
 *   pydem_twi                 DEMProcessor.calc_twi                pydem/dem_processing.py:1647-1677
 *
 * Ownership: the caller owns every host buffer it passes; the library owns device memory behind
 * the opaque pydem_tile handle (create / upload / run / download / destroy).  Nothing allocated
 * by the library is ever returned to the caller.  One HIP stream per handle; calls on different
 * handles may run from different host threads.
 */
#ifndef PYDEM_HIP_H
#define PYDEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pydem_tile pydem_tile;

/* device-resident per-tile fields, for pydem_tile_upload / pydem_tile_download */
enum pydem_field {
    PYDEM_ELEV = 0,        /* float64 [n,m]  conditioned elevation                              */
    PYDEM_MAG = 1,         /* float64 [n,m]  slope magnitude, -1 on flats                       */
    PYDEM_DIRECTION = 2,   /* float64 [n,m]  D-infinity direction (rad), -1 on flats            */
    PYDEM_FLATS = 3,       /* uint8   [n,m]  flats mask                                         */
    PYDEM_SECTION = 4,     /* int8    [n,m]  facet index 0..7, -1 on flats                      */
    PYDEM_PROPORTION = 5,  /* float64 [n,m]  share of flow to the facet's first neighbour       */
    PYDEM_UCA = 6,         /* float64 [n,m]  upstream contributing area, NaN on flats           */
    PYDEM_TWI = 7,         /* float64 [n,m]  ln(uca / (mag + min_slope)) (un-scaled)            */
    PYDEM_EDGE_TODO = 8,   /* uint8   [n,m]  inlet edge cells still waiting for a neighbour     */
    PYDEM_EDGE_DONE = 9,   /* uint8   [n,m]                                                    */
    PYDEM_FIELD_COUNT = 10
};

/* element types accepted by pydem_tile_upload for PYDEM_ELEV (held as float64 on the device; a PYDEM_F32 elevation
 * keeps the reference's float32 subtraction of elevations, dem_processing.py:1958-1962 / :1361) */
enum pydem_dtype { PYDEM_F64 = 0, PYDEM_F32 = 1, PYDEM_I16 = 2, PYDEM_I32 = 3, PYDEM_U8 = 4, PYDEM_I8 = 5 };

/* options of the hot path; names and defaults follow the DEMProcessor traits
 * (pydem/dem_processing.py:105-154) */
typedef struct pydem_options {
    int32_t drain_pits;              /* 1   :112 */
    int32_t drain_pits_min_border;   /* 0   :114 */
    int32_t drain_pits_max_iter;     /* 300 :117 */
    int32_t drain_pits_max_dist;     /* 32  :118 (0 = None) */
    double  drain_pits_max_dist_XY;  /* NaN = None :119 */
    int32_t apply_uca_limit_edges;   /* 0   :123 */
    int32_t apply_twi_limits;        /* 0   :125 */
    int32_t apply_twi_limits_on_uca; /* 0   :127 */
    int32_t circular_ref_maxcount;   /* 50  :151 */
    double  uca_saturation_limit;    /* 32  :146 */
    double  twi_min_slope;           /* 1e-3 :147 */
    double  twi_min_area;            /* +inf :148 (in/out: min(dX2*dY2) is folded in by pydem_uca) */
} pydem_options;

/* per-stage device time of the last call of each stage, milliseconds (hipEvent pairs) */
typedef struct pydem_timings {
    double slopes_directions_ms;  /* stencil + perimeter                      */
    double stencil_kernel_ms;     /* the interior 3x3 stencil kernel alone    */
    double flats_ms;              /* flats-edge labelling                     */
    double graph_ms;              /* section/proportion + pits + in-degree    */
    double pits_ms;               /* pit -> drain assignment alone            */
    double sweep_ms;              /* frontier sweep                           */
    double twi_ms;
    int64_t sweep_rounds;         /* frontier rounds of the last sweep        */
    int64_t sweep_kernel_launches;
    int64_t n_flats;              /* cells in the flats mask after slopes_directions */
    int64_t n_pit_edges;          /* pit -> drain edges built                  */
    int64_t n_pits_undrained;     /* the reference's "pits had no place to drain" count */
    int64_t n_unresolved;         /* cells still unfinished after the re-seed loop (circular drainage, dem_processing.py:951-964) */
    int64_t sweep_tile_passes;    /* LDS tile-local passes run before the queue rounds */
    int64_t n_pits;               /* pit candidates of the last pit search (all of them enter the lane pass)              */
    int64_t n_pits_row;           /* ... that entered the row pass (16 lanes per pit; 0: pass not run, PYDEM_PITS_ROW=0)  */
    int64_t n_pits_wave;          /* ... that entered the wavefront pass (128 x 128 window)                               */
    int64_t n_pits_big;           /* ... that entered the 256 x 256 wavefront pass or the workgroup pass behind it        */
} pydem_timings;

const char *pydem_hip_last_error(void);
int pydem_hip_device_count(int *count);
int pydem_hip_device_name(int device, char *buf, int buflen);
/* free / total bytes of the device's HBM (hipMemGetInfo): what a directory run sizes `tiles_in_flight` against, and the leak
 * check of the test suite (no counterpart in the reference, whose tiles live in host memory). */
int pydem_hip_device_memory(int device, int64_t *free_bytes, int64_t *total_bytes);

/* The conditioning stages (pydem_fill_flats, pydem_pit_candidates_read, pydem_pit_paths) lease one scratch arena per device
 * for the duration of a call (it grows to the largest request seen: ~6 GB + up to 17 GB for the large-window simulations
 * of a 8192 x 8192 tile); this returns every arena to the driver.  It also empties the per-device free lists on which
 * pydem_tile_destroy leaves the planes of a tile for the next tile of the same shape (PYDEM_PLANE_CACHE_GB, default 32;
 * a failing allocation empties them as well) and the pinned chunks / streams of the whole-plane transfers.  No counterpart in
 * the reference (host arrays). */
int pydem_hip_release_scratch(void);

int pydem_tile_create(int64_t n_rows, int64_t n_cols, int device, pydem_tile **out);
int pydem_tile_destroy(pydem_tile *t);
/* dX, dY: n_rows-1 values; dX2, dY2: n_rows values (DEMProcessor.__init__ :229-258) */
int pydem_tile_set_spacing(pydem_tile *t, const double *dX, const double *dY,
                           const double *dX2, const double *dY2);
int pydem_tile_upload(pydem_tile *t, int field, const void *src, int dtype);
int pydem_tile_download(pydem_tile *t, int field, void *dst);
/* one row (axis 0, n_cols elements) or one column (axis 1, n_rows elements) of a field, in the
 * field's own element type: the strips neighbouring tiles exchange in the directory flow
 * (reference pydem/process_manager.py:131-145 and :252-255 read/write them through zarr) */
int pydem_tile_get_line(pydem_tile *t, int field, int axis, int64_t index, void *dst);
int pydem_tile_set_line(pydem_tile *t, int field, int axis, int64_t index, const void *src);
/* `count` lines at once (fields[k], axes[k], indices[k] -> dsts[k]) with a single synchronisation: what one
 * calc_uca_ec of the reference reads from its neighbours' stores (process_manager.py:246-274) */
int pydem_tile_get_lines(pydem_tile *t, int count, const int *fields, const int *axes, const int64_t *indices,
                         void *const *dsts);
int pydem_tile_synchronize(pydem_tile *t);
int pydem_tile_timings(pydem_tile *t, pydem_timings *out);
int64_t pydem_tile_device_bytes(pydem_tile *t);

/* deterministic synthetic fractal DEM written straight into the tile's elevation
 * (bit-identical to pydem_amd/synth.py:fractal; bench / test input only) */
int pydem_tile_synth_fractal(pydem_tile *t, uint32_t seed, int64_t row0, int64_t col0,
                             int n_octaves, int top_shift, double zmin, double zrange);

/* Elevation conditioning on the resident elevation (csrc/cond_device.hip): DEMProcessor.calc_fill_flats
 * (pydem/dem_processing.py:551-579: quantisation artefacts :396-426 when max_pit_area > 0, then _fill_flat :308-394 for
 * every labelled flat), or only DEMProcessor.calc_fill_pit_artifacts with artefacts_only.  The elevation stays on the
 * device (float64 after the flats step, the values of the input dtype after the artefact step alone).  *needs_host = 1
 * and an untouched tile when the tile has no-data cells: the caller uses the host implementation
 * (pydem_cond_pit_artifacts / pydem_cond_fill_flats below) for those. */
int pydem_fill_flats(pydem_tile *t, double max_pit_area, int below_sea, double source_tol, int peaks, int pits, int artefacts_only,
                     int *needs_host);
/* DEMProcessor.calc_pit_drain_paths (pydem/dem_processing.py:428-548) on the resident float64 elevation
 * (csrc/cond_paths.hip).  Three calls because the processing order is numpy's: pydem_pit_candidates finds the strict local
 * minima (:444-449; *npits = -1 when the tile has no-data cells), pydem_pit_candidates_read returns their cells
 * (ascending) and elevations, the caller sorts them with np.argsort like the reference (:450-452 -- the tie order of
 * that call is part of the result) and pydem_pit_paths carves the paths in that order.  *needs_host = 1 (surface
 * restored) when the order-preserving parallel schedule had to give up; the host loop pydem_cond_pit_paths takes over. */
int pydem_pit_candidates(pydem_tile *t, int below_sea, int64_t *npits);
int pydem_pit_candidates_read(pydem_tile *t, int64_t npits, int32_t *cells, double *elev);
int pydem_pit_paths(pydem_tile *t, const int32_t *order, int64_t npits, int max_iter, int max_dist, double max_dist_XY,
                    int64_t *n_failed, int64_t *iter_used, int64_t *rounds, int *needs_host);
int pydem_slopes_directions(pydem_tile *t);
int pydem_find_flats(pydem_tile *t);
int pydem_uca(pydem_tile *t, pydem_options *opt);
/* the flow graph of pydem_uca (section / proportion / adjacency / pit edges, dem_processing.py:1021-1382) for a tile whose
 * elevation, slope, aspect and flats were uploaded instead of computed -- what the reference's edge worker rebuilds from
 * its stores before every round (process_manager.py:227-240) and a resumed directory job needs once; it resets the tile's
 * edge masks, so upload stored masks afterwards */
int pydem_build_graph(pydem_tile *t, pydem_options *opt);
/* strips in the order left, right, top, bottom; left/right have n_rows entries, top/bottom
 * n_cols; data = neighbour uca (+uca_edges), done/todo = uint8 (process_manager.py:252-255) */
int pydem_uca_edge_update(pydem_tile *t, pydem_options *opt,
                          const double *const data[4], const uint8_t *const done[4],
                          const uint8_t *const todo[4]);
/* The same edge-resolution step for the multi-worker schedule of ProcessManager.process_uca_edges
 * (pydem/process_manager.py:1214-1246, here deterministic waves): the state of the fix-up stays on the device
 * between rounds -- a cell is processed once, when the last unresolved inlet upstream of it resolves, instead of
 * once per round -- so a round costs the cells it finishes, not everything downstream of its seeds.  Masks after
 * a round (edge_todo / edge_done, :848-856) and the areas of finished cells are those of pydem_uca_edge_update;
 * areas of cells that are still downstream of an unresolved inlet lag behind until pydem_uca_edge_flush, which
 * hands them what their finished upstream cells hold (call it when the fix-up ends; pydem_uca_edge_update,
 * pydem_twi and downloads of PYDEM_UCA do it implicitly).  Not available with opt->apply_uca_limit_edges (-6). */
int pydem_uca_edge_round_inc(pydem_tile *t, pydem_options *opt,
                             const double *const data[4], const uint8_t *const done[4],
                             const uint8_t *const todo[4]);
/* the same round with the strips already in the tile's device buffers (written by pydem_board_eval) */
int pydem_uca_edge_round_inc_dev(pydem_tile *t, pydem_options *opt);
int pydem_uca_edge_flush(pydem_tile *t);
int pydem_twi(pydem_tile *t, pydem_options *opt);

/* The pit -> drain triplets built by the last pydem_uca (the reference's local pit_i, pit_j,
 * pit_prop, pydem/dem_processing.py:1378-1380), in emission order.  Call with src == NULL to get
 * the count. */
int pydem_tile_pit_edges(pydem_tile *t, int64_t *n, int32_t *src, int32_t *dst, double *w);

/* The flow graph the sweep runs on, one packed word per cell [n,m] -- the device's form of the adjacency matrix A of
 * _mk_adjacency_matrix (pydem/dem_processing.py:1072-1153; A is never materialised here): bits 0-7 = which of the 8
 * neighbours (NW N NE W E SW S SE) drain into the cell, bit 8 / 9 = the regular out-edge to the facet's first / second
 * neighbour survived the keep-filter (:1136-1137; weights = proportion, 1 - proportion), bit 10 / 11 = the cell has pit
 * out- / in-edges (pydem_tile_pit_edges), bits 12-14 = facet index.  For tests and debugging (the edge set can be held
 * against scipy's / the oracle's triplets). */
int pydem_tile_graph_words(pydem_tile *t, uint32_t *out);

/* Undo the slope patch of the drained pits (mag[pit] = -1 again, flats untouched).  In the
 * reference's directory flow the calc_uca worker never writes its patched slope back to the store
 * (pydem/process_manager.py:192-194), so later phases see slope == -1 at drained pits; the
 * ProcessManager drop-in calls this to keep that behaviour. */
int pydem_tile_restore_pit_slopes(pydem_tile *t);

/* Kernel-only timing hook for bench.py: runs the interior stencil `iters` times on the tile's
 * resident elevation and returns the average kernel time (ms) measured with hipEvents on the
 * tile's stream. */
int pydem_bench_stencil(pydem_tile *t, int iters, double *avg_ms);

/* ---- the reference's own native boundary, on a generic scipy CSC/CSR graph ----------------------
 * Same arguments and in-place behaviour as the Cython functions (host arrays in, updated in place):
 *   pydem_drain_area         cyutils.drain_area        pydem/cyfuncs/cyutils.pyx:78-116 (loop :119-187)
 *   pydem_drain_connections  cyutils.drain_connections pydem/cyfuncs/cyutils.pyx:35-46  (loop :49-72)
 * edge_todo / edge_todo_no_mask may be NULL.  The DEMProcessor entry points above do not use these
 * (they never materialise the matrix); they exist so code written against cyutils keeps working. */
int pydem_drain_area(double *area, uint8_t *done, uint8_t *ids, const int32_t *col_indptr, const int32_t *col_indices,
                     const double *col_data, const int32_t *row_indptr, const int32_t *row_indices, int64_t n_rows,
                     int64_t n_cols, double *edge_todo, double *edge_todo_no_mask, int skip_edge, int device);
int pydem_drain_connections(uint8_t *arr, uint8_t *ids, const int32_t *indptr, const int32_t *indices, int64_t n,
                            uint8_t set_to, int device);

/* ---- RCCL transport for the edge strips of the directory flow -------------------------------
 * Replaces the shared zarr store the reference's workers exchange strips through
 * (pydem/process_manager.py:243-255, write-verify-retry :362-381).  One communicator per process
 * (one process per GPU); rank 0 creates the 128-byte id and hands it to the other ranks by any
 * out-of-band channel.  A gather step is: begin(n) -> pack_line(...) for the lines this rank
 * owns -> allreduce(sum) -> every rank holds all lines (host copy in host_out). */
typedef struct pydem_comm pydem_comm;
int pydem_comm_unique_id(char *out128);
int pydem_comm_create(int world, int rank, const char *uid128, int device, pydem_comm **out);
int pydem_comm_destroy(pydem_comm *c);
int pydem_comm_begin(pydem_comm *c, int64_t n_doubles);
int pydem_comm_pack_line(pydem_comm *c, pydem_tile *t, int field, int axis, int64_t index, int64_t offset);
/* several lines of one tile without a host synchronisation (stream events order clear -> packs -> collective) */
int pydem_comm_pack_lines(pydem_comm *c, pydem_tile *t, int count, const int *fields, const int *axes, const int64_t *indices,
                          const int64_t *offsets);
int pydem_comm_put(pydem_comm *c, const double *host_in, int64_t n_doubles, int64_t offset);
int pydem_comm_allreduce(pydem_comm *c, int64_t n_doubles, int op /* 0 sum, 1 max */, double *host_out);

/* ---- edge board: the strips of ProcessManager.process_uca_edges stay on the device (comm.hip).
 * Replaces what the reference's edge worker and manager do on the host for every round: reading the neighbour
 * lines from the store, the corner rules and rule :274 (pydem/process_manager.py:243-274), the metrics
 * (calc_uca_ec_metrics :199-221).  The board holds every line some tile reads (doubles; masks as 0 / 1) at fixed
 * offsets, identically on every rank.
 *   pydem_board_set_desc  where tile `index` finds its lines: offsets28 = own_todo[4], own_done[4], nb_uca[4],
 *                         nb_done[4], nb_todo[4] (sides left, right, top, bottom), cnr_done[4], cnr_uca[4] (diagonal
 *                         pixel of the corners tl, tr, bl, br); -1 = missing; flags8 = nb_self[4] (the edge table
 *                         points the tile at its own line), cnr_1ov[4] (check_1overlap :286-293); `tile` = the
 *                         resident tile whose strip buffers the evaluation fills (NULL: a tile of another rank)
 *   pydem_board_set_lines which part of the board holds the lines of tile `index` (mb_start, size) and, for a tile of
 *                         this rank, the lines to gather from it (field / axis / index -> rel_offset inside that part); for a
 *                         tile of another rank (`tile` NULL) the same list gives the layout only (which lines are areas, which
 *                         masks: the queued waves carry masks as bytes through the collective)
 *   pydem_board_refresh   after a wave (the same tile list on every rank): the tiles of this rank gather their lines
 *                         into the wave staging buffer, one ncclAllReduce(sum) over disjoint fills when `c` is given,
 *                         then the staging buffer is copied into the board
 *   pydem_board_eval      strips (data, done, todo after the corner rules, rule :274 -- everywhere if full[k], else
 *                         only on the mosaic border -- and the adoption of finished neighbour values) into the
 *                         tiles' buffers, and 8 words per tile to `out` (all tiles of the board): cells 'todo' and
 *                         facing a finished neighbour, cells 'todo', 'todo' pixels the border rule / full rule :274
 *                         would drop, seeds, a hash of the strips */
typedef struct pydem_board pydem_board;
int pydem_board_create(int device, int n_tiles, int64_t n_doubles, pydem_board **out);
int pydem_board_destroy(pydem_board *b);
int pydem_board_set_desc(pydem_board *b, int index, int32_t n, int32_t m, const int64_t *offsets28, const int32_t *flags8,
                         pydem_tile *tile);
int pydem_board_set_lines(pydem_board *b, int index, int64_t mb_start, int64_t size, pydem_tile *tile, int count,
                          const int *fields, const int *axes, const int64_t *indices, const int64_t *rel_offsets);
int pydem_board_refresh(pydem_board *b, pydem_comm *c, int n_wave, const int *wave_tiles);
/* the same refresh with the sum over ranks done by the caller on the host (transports without RCCL: the torch.distributed
 * fallback, pydem_amd/parallel.py DistTransport): stage = zero + pack this rank's lines of the wave + copy to host_out
 * (*n_doubles values, at most cap); the caller sums the buffers of all ranks; unstage files the sum on the board.
 * Replaces the same store round trip as pydem_board_refresh (process_manager.py:243-255). */
int pydem_board_refresh_stage(pydem_board *b, int n_wave, const int *wave_tiles, double *host_out, int64_t cap, int64_t *n_doubles);
int pydem_board_refresh_unstage(pydem_board *b, int n_wave, const int *wave_tiles, const double *host_in, int64_t n_doubles);
int pydem_board_eval(pydem_board *b, int count, const int *tiles, const int *full, unsigned long long *out);
/* Up to k_waves (<= 64) waves of the same schedule queued back to back WITHOUT a look from the manager (the reference's manager
 * loop, pydem/process_manager.py:1214-1246, polls its workers and re-ranks after every completion; for a mosaic of at most
 * 2 * n_workers tiles the ranking :1177-1188 selects every tile with a positive metric, so the wave can be chosen by a
 * kernel): per wave a selection kernel (candidates = positive metric or 'todo' pixels dropped on the mosaic border, strips
 * changed since the tile's last round), for every tile of this rank a condensed round + the gather of its lines gated by the
 * wave's member word, one ncclAllReduce(sum, bytes) of the whole staging buffer (areas as doubles, masks as bytes) when `c` is
 * given, the unpacking onto the board and the
 * evaluation of the tiles that read the wave's lines.  `state`: 528 64-bit words in and out --
 *   [0] bit per tile: its rounds may be queued (a candidate without the bit stops the batch BEFORE its wave: the host runs
 *   that wave, e.g. a tile's first round, which builds its fix-up state), [1] out: 0 = all k_waves ran, 1 = the fix-up
 *   is over (no candidate, and no tile drops a 'todo' pixel under rule :274 everywhere -- while one does, the tile that
 *   drops the most runs alone with its strips evaluated under that rule: the tie-break wave of the schedule, also chosen by the
 *   kernel), 2 = a candidate needs the host, 3 = wave limit,
 *   [2] out: waves run, [3] waves allowed, [8+a] / [72+a] metric numerator / denominator of tile a as the schedule holds
 *   them (stale for diagonal neighbours like check_mets :1116-1136), [136+a] / [200+a] strip hash of a's last round / whether
 *   it has one, [264+a] bit mask of the tiles reading a line of a, [328+a] a and its four side neighbours, [392+w] out: the
 *   members of wave w, [7] out: the waves ran as captured hipGraphs (one launch per wave; default without a communicator, plain launches with one; PYDEM_EDGE_GRAPH=0 / 1 forces either),
 *   [456+a] scratch (round stamps), [521] out: tie-break waves of the batch, [522] out: bit w = wave w was one.
 *   `scal_out` as `out` of pydem_board_eval.  pydem_tile_edge_queue_ready: 1 when the tile's rounds can be queued (condensed fix-up state built by its first round, strips buffers attached by pydem_board_set_desc). */
int pydem_board_run_waves(pydem_board *b, pydem_comm *c, int k_waves, unsigned long long *state, unsigned long long *scal_out);
/* A batch contains collectives, so everything that can fail on ONE rank alone is a call of its own:
 *   pydem_board_prepare_waves  checks / builds what a batch over the tiles `ok_tiles` (bit per tile, = state[0]) needs -- the
 *       tiles' fix-up state, watched lines, the staging layout (`staged` != 0: the batch sums its staging buffer over the ranks)
 *       and the device tables -- and enqueues nothing.  With several ranks the caller lets the ranks agree on its verdict
 *       (one allreduce) before any of them calls pydem_board_run_waves; that call then only enqueues.  (A single process may
 *       skip it: run_waves prepares what is not prepared.)
 *   pydem_board_run_waves_ex   the same batch with the sum over the ranks done by the CALLER (`exchange`, not RCCL: processes
 *       sharing one GPU, transports without RCCL): per wave the staging buffer goes to the host, exchange(ctx, 0, bytes, n)
 *       sums it byte-wise in place, and -- what RCCL cannot do -- exchange(ctx, 1, doubles, 8) (maximum in place) first
 *       checks that every rank selected the SAME wave (-8 and a message otherwise).  One host look per wave: the tested
 *       restatement of the queued path (every rank runs the selection kernel from replicated numbers), not the fast one.
 * A batch that does not come back within PYDEM_EDGE_TIMEOUT seconds (default 300, 0 = wait for ever) returns -7 with the last
 * wave selected on this rank in the message and aborts the communicator (ncclCommAbort): ranks whose schedules disagree
 * would otherwise sit in ncclAllReduce without a word.  Replaces the manager's poll loop, process_manager.py:1214-1246.
 *   pydem_comm_count           ranks behind the communicator (ncclCommCount) */
typedef int (*pydem_exchange_fn)(void *ctx, int op, void *buf, int64_t n);
int pydem_board_prepare_waves(pydem_board *b, int staged, unsigned long long ok_tiles);
int pydem_board_run_waves_ex(pydem_board *b, pydem_comm *c, int k_waves, unsigned long long *state, unsigned long long *scal_out,
                             pydem_exchange_fn exchange, void *ctx);
int pydem_comm_count(pydem_comm *c, int *count);
int pydem_tile_edge_queue_ready(pydem_tile *t);
int pydem_board_download(pydem_board *b, double *out);

/* ---- elevation conditioning: host-side inner loops (no device work; pydem_amd/conditioning.py keeps the
 * vectorised prologues -- 3x3 filters, scipy.ndimage.label, the numpy argsort whose tie order is part of the
 * result -- and calls these for the per-region / per-pit loops of the reference)
 *   pydem_cond_pit_artifacts: calc_fill_pit_artifacts (dem_processing.py:396-426); lab = labels (1..nlab) of the
 *       candidate depressions, raise[c] = 1 where the cell is lifted by one unit
 *   pydem_cond_fill_flats:    _fill_flat (:308-394) for every labelled flat of calc_fill_flats (:551-579);
 *       data = unmodified surface (float64, NaN = masked), built = copy of it that receives the new flats
 *   pydem_cond_pit_paths:     calc_pit_drain_paths (:428-548) for the pits in the given order, surface edited in place */
int pydem_cond_pit_artifacts(const double *elev, int64_t n_rows, int64_t n_cols, const int32_t *lab, int32_t nlab,
                             double max_area, int f32 /* the surface holds float32 values */, uint8_t *raise);
int pydem_cond_fill_flats(const double *data, double *built, int64_t n_rows, int64_t n_cols, const int32_t *lab,
                          int32_t nlab, double source_tol, int peaks, int pits);
int pydem_cond_pit_paths(double *elev, int64_t n_rows, int64_t n_cols, const int64_t *pits, int64_t npits,
                         const double *dX, int64_t n_dX, const double *dY, int max_iter, int max_dist,
                         double max_dist_XY, int dtype_mode /* 0 float64, 1 integer, 2 float32 surface (values as float64) */,
                         int64_t *n_failed, int64_t *iter_used);

/* TIFF LZW encoder for the GeoTIFF export (host code): the reference writes its exports with rasterio's compress='lzw'
 * (pydem/process_manager.py:905, :930).  dst must hold up to n * 3 / 2 + 16 bytes (incompressible data grows by 12 %);
 * *out_n = bytes written.  The stream is libtiff's for the same input. */
int pydem_tiff_lzw_encode(const uint8_t *src, int64_t n, uint8_t *dst, int64_t cap, int64_t *out_n);

#ifdef __cplusplus
}
#endif
#endif /* PYDEM_HIP_H */
