/*
 * lbmdem_hip.h -- C ABI of the MI355X-native 2D LBM-DEM hot path (liblbmdem_hip.so).
 *
 * Drop-in boundary. The reference (cb-geo/2d-lbm-dem, src/main.c) has no plugin/FFI interface:
 * its seam is the set of `void fn(void)` routines that renderScene() (main.c:1697-1777) calls on
 * file-scope globals. Each entry point below replaces one of those call sites; the reference line
 * it stands in for is cited next to it. INTEGRATION.md shows the edit a maintainer of the reference
 * would make to main.c to call this library instead of its own loops.
 *
 * Conventions: plain C types only; every function returns 0 on success and a negative
 * LBMDEM_E* code on failure (lbmdem_last_error() gives the text); no exceptions cross the ABI;
 * the library owns all device memory, the caller owns every host buffer; one host thread per
 * handle. Host-side lattice data uses the REFERENCE layout f[x][y][q], x slow (main.c:56,1802);
 * the device layout (tiles of 16 consecutive y with the nine directions of a tile contiguous,
 * f[x][y / 16][q][y % 16]) is internal.
 *
 * There is no CPU fallback: every entry point fails with LBMDEM_ENODEVICE when no HIP device
 * is usable.
 */
#ifndef LBMDEM_HIP_H
#define LBMDEM_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The reference fixes its lattice with the object-like MACROS lx, ly and scale (main.c:24-32; its benchmark passes
 * them as -Dlx= -Dly= -Dscale=, benchmark.xml:85). They would rewrite the member and parameter names below, so they
 * are parked while this header is read and restored at its end: the header can be included anywhere in the reference's
 * main.c (found by compiling the binding of INTEGRATION.md: oracle/make_integration_check.py). */
#pragma push_macro("lx")
#pragma push_macro("ly")
#pragma push_macro("scale")
#undef lx
#undef ly
#undef scale

#define LBMDEM_OK 0
#define LBMDEM_EINVAL (-1)    /* bad argument / bad state */
#define LBMDEM_ENODEVICE (-2) /* no usable HIP device */
#define LBMDEM_EHIP (-3)      /* a HIP runtime call failed */
#define LBMDEM_ENOMEM (-4)

typedef struct lbmdem_handle lbmdem_handle;

/* Physics constants: the initialised globals of main.c:74-118,143,163-165. */
typedef struct lbmdem_physics {
  double rho_moy, tau, s2, s3, s5, s7, s8, s9, nu, reductionR; /* main.c:74-94  */
  double G, angleG;                                            /* main.c:97-98  */
  double km, kg, kt, ktm, nug, num, nugt;                      /* main.c:104-109 */
  double mu, mum, mumb, murf;                                  /* main.c:110-113 */
  double distVerlet, dtt, iterDEM;                             /* main.c:115-118 */
  double freq, amp, t;                                         /* main.c:163-165 */
  int updateVerlet;                                            /* main.c:116 */
  int stepFilm;                                                /* main.c:143 */
} lbmdem_physics;

/* Run configuration. The derived block is what main() computes at main.c:1836-1860; fill it with
 * lbmdem_derive() (bit-identical arithmetic) or by hand. */
typedef struct lbmdem_config {
  int lx, ly;         /* global lattice (reference macros lx, ly, main.c:27-32) */
  int x_begin, x_end; /* owned strip [x_begin, x_end) of the global lattice; 0, lx on one GPU */
  int halo;           /* extra rows kept beyond each interior cut (0 on one GPU) */
  int device;         /* HIP device ordinal */
  int nbgrains;
  double scale;       /* reference macro `scale` (main.c:24-26) */
  /* derived (main.c:1836-1860) */
  double dx, dtLB, c, dt, dt2;
  int npDEM;
  double Mgx, Mdx, Mby, Mhy; /* wall positions (main.c:201-204,1836-1839) */
  double xG, yG;             /* gravity components (main.c:1841-1842) */
  lbmdem_physics phys;
} lbmdem_config;

/* defaults = the reference's initialisers */
int lbmdem_physics_defaults(lbmdem_physics* p);

/* Time-step derivation, main.c:1836-1860: dx, dtLB, dtmax, npDEM, c, dt from lx, ly, scale and the
 * smallest radius (r in metres). Also sets walls and gravity. Uses cfg->phys. */
int lbmdem_derive(lbmdem_config* cfg, int lx, int ly, double scale, int nbgrains, const double* r);

/* Sample reader, main.c:609-639 (read_sample): comment line, count, then count x "r x y[;]" in
 * units of 1 mm. Returns metres. Arrays are malloc'ed; release with lbmdem_free_host(). */
int lbmdem_read_sample(const char* path, int* nbgrains, double** r, double** x1, double** x2);
void lbmdem_free_host(void* p);

/* Allocation + initial state, replacing main.c:1802-1834,1858-1861: f = w[q] (init_density,
 * main.c:716-724), grains at rest with m, It from r (main.c:624-635), rLB, and the initial obstacle
 * map (init_obst, main.c:663-711). r, x1, x2 in metres, nbgrains entries each. */
int lbmdem_create(const lbmdem_config* cfg, const double* r, const double* x1, const double* x2,
                  lbmdem_handle** out);
int lbmdem_destroy(lbmdem_handle* h);

/* ---- the hot path --------------------------------------------------------------------------- */

/* One fluid step = main.c:1711-1713,1717: reinit_obst_density (966-986), obst_construction
 * (991-1065), collision_streaming (1071-1243), forces_fluid (1285-1333). */
int lbmdem_lbm_step(lbmdem_handle* h);
/* The same, phase by phase (lbm_step == obst_construction; collide_stream; forces_fluid).
 * reinit_obst_density is folded into collide_stream: it needs the *previous* obstacle map, which
 * the library keeps. */
int lbmdem_obst_construction(lbmdem_handle* h); /* main.c:991-1065 (obst only; act, delta are recomputed on the fly) */
int lbmdem_collide_stream(lbmdem_handle* h);    /* main.c:966-986 + 1071-1243 */
int lbmdem_forces_fluid(lbmdem_handle* h);      /* main.c:1285-1333 */

/* Diagnostic for the parity force kernel: the fused collide_stream kernel leaves every bounce-back link's
 * momentum-exchange sum (main.c:1313-1316) in a per-grain table, and forces_fluid replays them in the reference's
 * order without touching the lattice; grains with a link that ends in a non-fluid node (another grain, a
 * lattice-edge wall), with overlapping discs, or cut by a strip boundary are gathered from obst and f instead --
 * same bits either way. Returns how many grains of the LAST forces_fluid call took each route (from_table = 0
 * when the table did not describe the current lattice and every grain was gathered). Synchronises. */
int lbmdem_force_stats(lbmdem_handle* h, int* from_table, int* gathered);
/* Which size-dependent fast paths are active on this handle: info4[0] link-sum table (needs < 2^18 grains, reduced radius
 * < ~20 nodes, reductionR < 1), info4[1] its slots per direction, info4[2] the lowest-cover record that keeps `act` exact
 * where three or more reduced discs overlap (< 2^20 grains; else the two-disc rule), info4[3] the marching fused kernel. */
int lbmdem_path_info(lbmdem_handle* h, int* info4);
/* How the fused kernel's launch over this handle's rows is cut into work items (windows of 60 columns -- MARCH_WW -- x segments of rows):
 * info12 = {levels, rows per XCD band, rows per interleaved chunk, segment rows of level 0..3, band rows cut at level 0..3,
 * work items in all}. levels == 0: uniform segments of info12[3] rows (short row ranges: strips, small lattices); else the
 * tapered order (long segments first, short ones last) of DESIGN.md section 4. Host-side arithmetic only. */
int lbmdem_fused_work_order(lbmdem_handle* h, int* info12);

/* initVerlet + VerletWall, main.c:1519-1594 (same pair set; uniform grid + radix sort instead of
 * the O(N^2) scan). Also moves the right/top DEM walls as VerletWall does (main.c:1555-1561). */
int lbmdem_verlet_rebuild(lbmdem_handle* h);

/* One DEM sub-step = main.c:1733-1764: drift + half kick, acceleration_grains (1336-1516, film
 * law when nbsteps % stepFilm == 0), second half kick, nbsteps++. */
int lbmdem_dem_substep(lbmdem_handle* h);

/* n x renderScene() (main.c:1697-1765) with the reference's cadences: a fluid step when
 * nbsteps % npDEM == 0, a Verlet rebuild when nbsteps % updateVerlet == 0, then a DEM sub-step.
 * Device-resident; returns without synchronising. */
int lbmdem_run(lbmdem_handle* h, long n_dem_steps);
/* The same without the fluid steps: n x (Verlet rebuild when due; DEM sub-step). For drivers that run the
 * fluid step themselves (strip decomposition: halo exchange and force combine sit between its phases). */
int lbmdem_run_dem(lbmdem_handle* h, long n_dem_steps);

/* EXTENSION, not in the reference as it runs: a lid. The reference's top-plate copies carry commented-out moving-wall
 * terms (main.c:1129-1130: f[x][ly-1][3] = f[x-1][ly-2][7]; //-uw_h/6;  f[x][ly-1][5] = f[x+1][ly-2][1]; //+uw_h/6;
 * `uw_h` is not even declared). lbmdem_set_lid enables exactly those two terms with uw_h in lattice units (0 = off, the
 * default) -- BASELINE.json configs[1], a lid-driven cavity. Checked against the CPU oracle carrying the same terms. */
int lbmdem_set_lid(lbmdem_handle* h, double uw_h);

/* hydrodynamic-force summation: 0 = parity (the reference's order of additions, bit-exact); 1 = fast (the same
 * addends reduced across the lanes of a wavefront: differs in the last bits; same speed as parity since round 2,
 * kept for callers that do not need the reference's bits). Default 0. */
int lbmdem_set_force_mode(lbmdem_handle* h, int mode);

/* lbmdem_run / lbmdem_run_dem / lbmdem_comm_run hand every run of ordinary sub-steps (renderScene calls between which
 * nothing else happens: no fluid step main.c:1710, no list rebuild main.c:1721, regular contact law main.c:1427-1451, no
 * write_DEM diagnostics main.c:1773) to ONE kernel launch of at most `max_substeps` sub-steps: the tiles of grains hand
 * their drifted state (main.c:1748-1753) to their partners' tiles through tagged cache-line records instead of kernel
 * boundaries. Same bits as lbmdem_dem_substep called that many times. < 2: one launch per sub-step. Default 128. The
 * library falls back to one launch per sub-step by itself where the tiles of a packing cannot all be resident at once. */
/* obst_construction (main.c:991-1065) clears the map and paints every disc again; between two fluid steps a disc moves by
 * a fraction of a node. on = 1: the rasteriser compares every disc's footprint at the centre it was last painted at in that
 * map buffer with its footprint now and writes only the nodes whose owner changes (no reset of the canvas; a disc that has
 * not moved far enough for any node to change sides is skipped altogether; needs the pair list: falls back to clear +
 * repaint by itself before the first list, after an upload of positions, with distributed grains, and while a grain has
 * outrun the list). Same maps bit for bit. on = 0 (default): clear + repaint every step -- on an agitated packing the update
 * is no faster (the rasteriser is bound by its per-grain set-up, not by its stores: 36 against 30 + 4 us at 50 000 grains), on
 * one at rest it is (16 us). lbmdem_obst_stats: how often each ran. */
int lbmdem_set_obst_update(lbmdem_handle* h, int on);
int lbmdem_obst_stats(lbmdem_handle* h, long* updates, long* repaints);
/* reinit_obst_density (main.c:966-986) needs the PREVIOUS owner of a node, i.e. the map of the step before, at the few
 * thousand nodes that changed hands. mode 1 (default): a rasterisation in place by the end of a run of sub-steps also
 * leaves one bit per lattice row and 64-column window of the fused kernel -- "the two maps differ here" -- and the fused
 * kernel of a whole single-domain step reads the second map only in those rows (4 of its 152 bytes per node otherwise).
 * 0: both maps are read everywhere. 2: as 1, and every use is checked (tests): the bits against the two maps, and the
 * populations against a second launch that reads both maps everywhere (into a scratch lattice).
 * lbmdem_change_mask_stats: fused launches that used the bits; what mode 2 found wrong -- low 32 bits: (row, window)
 * pairs whose bit was clear over differing maps, high 32 bits: populations that differed -- must be 0. */
int lbmdem_set_change_mask(lbmdem_handle* h, int mode);
int lbmdem_change_mask_stats(lbmdem_handle* h, long* used, long* hidden);

int lbmdem_set_dem_chain(lbmdem_handle* h, int max_substeps);
/* what that path has done so far: launches, sub-steps they covered, the workgroups ("tile slots": 64 grains each) one
 * launch needs resident at once, and how many the census found resident (-1: not taken yet, 0: they do not fit) */
int lbmdem_dem_chain_stats(lbmdem_handle* h, long* launches, long* substeps, int* tile_slots, int* resident);
/* A run that ends where a fluid step begins also rasterises the reduced discs (obst_construction's paint, main.c:1009-1032)
 * at the positions it ends with -- they and those of all partners are in the tiles' on-chip memory then: the next
 * lbmdem_obst_construction has nothing left to launch. lbmdem_dem_chain_paints: how often that happened.
 * lbmdem_set_dem_chain(h, -1) switches only this off (A/B). Same maps bit for bit. */
int lbmdem_dem_chain_paints(lbmdem_handle* h, long* paints);
/* A launch of that kernel needs all its workgroups on the GPU at once, which HIP does not promise (another process, a CU
 * mask): every wait in it is bounded, and a launch that gives up cannot end a run -- the reference's loop (main.c:1733-1763)
 * cannot fail either. It raises a stop word that every kernel queued behind it looks at first, so the device keeps the
 * state the launch started from; the next call that is not lbmdem_run / lbmdem_run_dem (or the 256th launch since) drains
 * the stream, takes the handle back to that launch, switches the multi-sub-step kernel off for the handle and repeats the
 * sub-steps since, one launch each: the same bits. lbmdem_dem_chain_recoveries: how often that has happened (0 on a GPU of
 * its own). Distributed handles (lbmdem_dist_*) report the failure instead: a rank cannot go back alone. */
int lbmdem_dem_chain_recoveries(lbmdem_handle* h, long* count);
/* The grains of that kernel's workgroups ("tiles" of 64). mode 1 (default): consecutive stretches along a space-filling
 * curve over the positions at lbmdem_create (or at this call) -- compact patches of the packing whatever the numbering of the
 * grains, so that a tile's partners outside itself are the patch's rim: the reference's own bin/50000.data, whose numbering
 * is not coherent in space, then runs at the pace of a row-numbered packing. 0: by index (tile t = grains 64 t .. 64 t + 63).
 * Only speed depends on it: a grain's sums keep the reference's order (partners ascending by index, main.c:1427-1451). */
int lbmdem_set_dem_tiles(lbmdem_handle* h, int mode);

/* ---- state in / out (host layout) ------------------------------------------------------------ */

int lbmdem_upload_f(lbmdem_handle* h, const double* f_aos);   /* [lx][ly][9]; rows of the local strip+halo are read */
int lbmdem_download_f(lbmdem_handle* h, double* f_aos);       /* [lx][ly][9]; only the OWNED rows are written */
int lbmdem_download_obst(lbmdem_handle* h, int* obst);        /* [lx][ly]; owned rows */
int lbmdem_download_macro(lbmdem_handle* h, double* rho, double* ux, double* uy); /* [lx][ly] each; owned rows; sums of f, f*ex, f*ey as write_vtk forms them (main.c:315-319) */
int lbmdem_total_density(lbmdem_handle* h, double* sum);      /* sum of f over the owned rows, tree order (fast; last bits differ from the reference's serial sum) */
/* check_density / final_density (main.c:1249-1273) with the reference's own bits: the serial chain sum = sum + f[x][y][q]
 * (x outer, y, q inner) continued from `sum_in` over the owned rows -- 0 for one domain; a strip passes its result to the
 * next strip. Computed on the device as integer quanta per lattice row wherever the running sum stays in one binade
 * (exact), rows where it does not are replayed element by element; `rows_replayed` (may be null) counts those. */
int lbmdem_total_density_serial(lbmdem_handle* h, double sum_in, double* sum_out, int* rows_replayed);
/* kinematics table, 9 doubles per grain: x1 x2 x3 v1 v2 v3 a1 a2 a3 */
int lbmdem_upload_kinematics(lbmdem_handle* h, const double* k9);
int lbmdem_download_kinematics(lbmdem_handle* h, double* k9);
int lbmdem_download_fhf(lbmdem_handle* h, double* fhf3);      /* interleaved fhf1,fhf2,fhf3 (main.c:180) */
/* Verlet lists in the reference's form: cumul[n] (main.c:150,1539), neighbours (j > i, ascending),
 * wall membership flags per grain (bit0 B, bit1 T, bit2 L, bit3 R; main.c:1563-1593).
 * `cap` = capacity of neighbours[]; *npairs receives the pair count (call with cap = 0 to size). */
int lbmdem_download_verlet(lbmdem_handle* h, int* cumul, int* neighbours, int cap, int* npairs,
                           int* wallflags);
/* grain pressure g.p of the last DEM sub-step (main.c:187: sum of the normal contact forces) */
int lbmdem_download_grain_pressure(lbmdem_handle* h, double* p);
/* The five float32 fields write_vtk builds (main.c:272-323): grain_pressure[ly][lx], grain_velocity
 * [ly][lx][3], grain_acceleration[ly][lx][3], fluid_pressure[ly][lx], fluid_velocity[ly][lx][3] (owned
 * rows: lx = x_end - x_begin); computed on the device with the reference's float accumulation. */
int lbmdem_download_vtk_fields(lbmdem_handle* h, float* grain_pressure, float* grain_velocity,
                               float* grain_acceleration, float* fluid_pressure, float* fluid_velocity);
/* write_vtk (main.c:237-338 -> visit_writer.c write_rectilinear_mesh, binary): writes the five files
 * <dir>/{grain_pressure,grain_velocity,grain_acceleration,fluid_pressure,fluid_velocity}_NNNNNN.vtk,
 * byte-identical to the reference's. Single-domain handles only. */
int lbmdem_write_vtk(lbmdem_handle* h, const char* dir, int nfile);
/* Contact diagnostics of the last DEM sub-step (what write_DEM prints, main.c:340-438). They are produced
 * in the sub-step that brings the step counter to a multiple of stepStrob = 4000 (main.c:142,1773), or in
 * every sub-step after lbmdem_set_diagnostics(h, 1). Table: 30 doubles per grain in the reference's struct
 * order (main.c:182-197): x1 x2 x3 v1 v2 v3 a1 a2 a3 r m mw It p s f1 f2 ifm fm fr ifr M11 M12 M21 M22 ice
 * slip rw z zz. fr, ice, slip, rw read "previous contact" carries that thread through the reference's serial
 * contact loop and from sub-step to sub-step (pft, pff, pf, ic: main.c:130-131); the library replays them in
 * the reference's order in the table sub-steps; the carries such a sub-step starts from are those of the last
 * contact of each kind however many sub-steps ago (every ordinary sub-step records its last contacts per tile of
 * grains; single-domain handles). */
int lbmdem_set_diagnostics(lbmdem_handle* h, int always);
int lbmdem_download_grain_table(lbmdem_handle* h, double* table30);
/* write_DEM (main.c:340-438): <dir>/DEM%06d.dat and one line appended to <dir>/stats.data; energies8 (may be
 * NULL) receives energie_cin, energy_p, SE, IFR, WF, INCE, TSLIP, TRW for the console line of main.c:1885-1889. */
int lbmdem_write_dem(lbmdem_handle* h, const char* dir, int nfile, double* energies8);
/* write_forces (main.c:440-478): <dir>/DEM%06d.ps, a PostScript picture of the grains (grey level from g.fm)
 * and one line per overlapping pair, in the reference's order (i outer, j inner, both directions). The
 * reference reads g[nbgrains], one element past its array, and its three "%%%Word" header formats are
 * undefined conversions; this writer emits the nbgrains real grains and the headers the format strings
 * evidently mean. Every other line is character-identical to the reference's file. The pair search is a
 * host-side uniform grid (same pairs as the reference's O(N^2) loop). */
int lbmdem_write_forces(lbmdem_handle* h, const char* dir, int nfile);
/* Checkpoint / restart (absent in the reference, which cannot resume a run: SURVEY.md section 5). The file
 * holds exactly the state that defines the continuation at a renderScene() boundary -- populations,
 * current obstacle map, grain kinematics, hydrodynamic forces, Verlet lists, wall positions, step
 * counter, the force-kernel choice, the diagnostics switch and the "previous contact" carries of the
 * order-dependent diagnostics -- in the device layout of the strip that wrote it (a layout word in the
 * header rejects files of another layout). A run restarted from it is bit-identical to the uninterrupted
 * run. load creates a new handle on `device`. */
int lbmdem_checkpoint_save(lbmdem_handle* h, const char* path);
int lbmdem_checkpoint_load(const char* path, int device, lbmdem_handle** out);
long lbmdem_nbsteps(lbmdem_handle* h);
int lbmdem_set_nbsteps(lbmdem_handle* h, long n);
int lbmdem_get_config(lbmdem_handle* h, lbmdem_config* out);

/* ---- streams, timing, multi-GPU plumbing ------------------------------------------------------ */

/* Enqueue on a caller-owned hipStream_t (NULL is the HIP default stream -- what PyTorch calls its
 * default stream), e.g. so that work is ordered with torch.distributed collectives; the library's
 * private non-blocking stream is the default and lbmdem_use_own_stream() goes back to it. */
int lbmdem_set_stream(lbmdem_handle* h, void* hip_stream);
int lbmdem_use_own_stream(lbmdem_handle* h);
int lbmdem_sync(lbmdem_handle* h);
/* HIP-event timing of the dominant kernel (fused collide+stream), on the stream it is launched on.
 * enable, run steps, then read the mean duration and the launch count. on = N > 1: every N-th launch is timed (the two
 * event records around a launch hold the next dispatch back: ~10 us per coupled step when every launch is timed). */
int lbmdem_profile_enable(lbmdem_handle* h, int on);
int lbmdem_profile_read(lbmdem_handle* h, double* mean_ms, long* launches);
/* GB/s (bytes read + bytes written) of a plain copy kernel moving `bytes` on this handle's device and stream: the best of
 * four shapes (8 or 16 bytes per lane, 2 048 to 8 192 workgroups), each the best of `reps` passes after a warm-up: the
 * yardstick bench.py puts next to the fused kernel's traffic rate (boxes differ by +-5 %). */
int lbmdem_measure_copy(lbmdem_handle* h, size_t bytes, int reps, double* gb_per_s);

/* Strip decomposition along x (one process per GPU); halo >= 2 rows (with REPLICATED grains, i.e. without
 * lbmdem_dist_enable, halo >= 2 + the largest grain radius in nodes). After collide_stream the `halo` outermost
 * OWNED rows on each interior side are packed into a caller-provided DEVICE buffer
 * (9 * halo * ly doubles, plane-major), exchanged by the caller (RCCL send/recv via
 * torch.distributed) and unpacked into the neighbour's halo rows. side: 0 = low x, 1 = high x. */
long lbmdem_halo_doubles(lbmdem_handle* h);
/* collide_stream in two parts, so that the halo exchange overlaps the bulk of the kernel:
 *   LBMDEM_CS_EDGES     the owned rows within `halo` of an interior cut (what the neighbours need); the
 *                       new lattice becomes current, halo_pack may follow immediately;
 *   LBMDEM_CS_INTERIOR  the remaining owned rows. Must follow EDGES before anything else reads the
 *                       lattice (forces_fluid, downloads, the next collide_stream fail until then).
 * EDGES + INTERIOR == lbmdem_collide_stream, row for row (every row is computed from the old lattice). */
#define LBMDEM_CS_EDGES 1
#define LBMDEM_CS_INTERIOR 2
int lbmdem_collide_stream_part(lbmdem_handle* h, int part);
int lbmdem_halo_pack(lbmdem_handle* h, int side, void* dev_buf);
int lbmdem_halo_unpack(lbmdem_handle* h, int side, const void* dev_buf);
int lbmdem_halo_pack2(lbmdem_handle* h, void* buf_lo, void* buf_hi);   /* both sides in one launch; null skips a side */
int lbmdem_halo_unpack2(lbmdem_handle* h, const void* buf_lo, const void* buf_hi);
/* ---- strips with the GRAINS distributed over the ranks (no collective) ------------------------------------------
 * Every rank keeps arrays for all grains (global index = array index) but integrates only the grains whose centre
 * lies in its rows plus a margin of `margin_rows` on either side, deep enough that what it does not integrate cannot
 * influence an owned grain within the npDEM sub-steps between two fluid steps (one Verlet-list edge per sub-step).
 * Per fluid step, with both neighbours, three point-to-point messages (fixed capacities, device buffers):
 *   LBMDEM_MSG_KIN     kinematics of the owned grains within the neighbour's margin -- also how a grain that crossed
 *                      the cut changes owner; may travel while the fluid step runs, unpack before the sub-steps;
 *   LBMDEM_MSG_TABLES  this rank's part of the link-sum tables of grains the neighbour owns (packed after
 *                      collide_stream and the f halo exchange, unpacked by the owner before forces_fluid);
 *   LBMDEM_MSG_FHF     hydrodynamic forces of the grains of the KIN message (after forces_fluid, before the sub-steps).
 * Sequence of one period: dist_begin_period; pack KIN (both sides); obst_construction; collide_stream (or its two
 * parts + halo exchange); pack TABLES / exchange / unpack TABLES; forces_fluid; unpack KIN; pack FHF / exchange /
 * unpack FHF; run_dem(npDEM). Results equal the single-GPU run bit for bit for any number of strips.
 * Needs halo >= 2 only. Overlapping reduced discs across a cut, or more grains near a cut than the message
 * capacities, are reported by lbmdem_sync. write_DEM's order-dependent diagnostics are not available in this mode. */
#define LBMDEM_MSG_KIN 0
#define LBMDEM_MSG_FHF 1
#define LBMDEM_MSG_TABLES 2
int lbmdem_dist_default_margin(lbmdem_handle* h);          /* rows */
int lbmdem_dist_margin_for(const lbmdem_config* cfg, double rmax_m);   /* the same from a derived config and the largest radius (host only) */
int lbmdem_dist_enable(lbmdem_handle* h, int margin_rows); /* 0 = default; the strip must be at least that wide */
long lbmdem_dist_message_doubles(lbmdem_handle* h, int kind); /* capacity of one message, in doubles */
int lbmdem_dist_begin_period(lbmdem_handle* h);
int lbmdem_dist_pack(lbmdem_handle* h, int kind, int side, void* dev_buf);
int lbmdem_dist_unpack(lbmdem_handle* h, int kind, int side, const void* dev_buf);
/* both sides in one launch (a null buffer skips the side): fewer dependent kernel launches per fluid step */
int lbmdem_dist_pack2(lbmdem_handle* h, int kind, void* buf_lo, void* buf_hi);
int lbmdem_dist_unpack2(lbmdem_handle* h, int kind, const void* buf_lo, const void* buf_hi);

/* ---- RCCL transport of that protocol for a C host (the driver 2d-lbm-dem_amd/host/lbmdem --gpus N); strips.py does
 * the same over torch.distributed. One process per GPU; rank k talks to ranks k-1 and k+1 only: ncclSend / ncclRecv
 * grouped per message class on side streams (the kinematics and the f halo rows travel while kernels run), no
 * collective on the step path. RCCL is dlopen'ed by the first of these calls. */
typedef struct lbmdem_comm lbmdem_comm;
#define LBMDEM_COMM_ID_BYTES 512          /* four RCCL unique ids: one communicator per message class */
int lbmdem_comm_unique_id(void* id);      /* rank 0 makes it, every rank passes the same bytes to ..._create */
int lbmdem_comm_create(const void* id, int rank, int world, int device, lbmdem_comm** out);
int lbmdem_comm_destroy(lbmdem_comm* c);
/* one fluid step of a handle in distributed-grain mode with its neighbours (the sequence documented above) */
int lbmdem_comm_lbm_step(lbmdem_handle* h, lbmdem_comm* c);
/* n x renderScene() (main.c:1697-1765) with that fluid step */
int lbmdem_comm_run(lbmdem_handle* h, lbmdem_comm* c, long n_dem_steps);
/* Drop-in outputs of a strip decomposition (not on the step path; output cadence).
 * The sub-step that feeds write_DEM -- the one that brings the step counter to a multiple of 4000, main.c:1773 -- is
 * run by ONE rank on a full replica, because four columns of the table (fr, ice, slip, rw) thread "previous contact"
 * carries through all contacts in grain-index order (main.c:130-131):
 *   lbmdem_dist_export_owned   state12 [n][12] (9 kinematic columns + fhf1..3) of the grains this rank owns, zeros
 *                              elsewhere; owned [n]; per carry the youngest record among the owned grains' contacts:
 *                              carry_keys [3][2] ({0,0} = none since the last table sub-step), carry_vals [3]
 *   (the caller merges the exports: disjoint, every grain has one owner; per carry the greatest key pair wins)
 *   lbmdem_dist_table_substep  the root imports the merged state, rebuilds its Verlet list from it and runs the
 *                              sub-step for all n grains with the single-domain diagnostic pipeline; afterwards
 *                              lbmdem_download_grain_table / lbmdem_write_dem / lbmdem_write_forces work on it.
 * lbmdem_comm_run does all of this over RCCL (rank 0 = root). write_vtk: lbmdem_vtk_place_owned drops the owned
 * columns into zeroed lattice-sized arrays (fields11: grain_pressure[cnt], grain_velocity[3 cnt],
 * grain_acceleration[3 cnt], fluid_pressure[cnt], fluid_velocity[3 cnt], each [ly][lx]), the merged arrays go to
 * lbmdem_write_vtk_fields; lbmdem_comm_write_vtk = both over RCCL, rank 0 writes. */
int lbmdem_dist_export_owned(lbmdem_handle* h, double* state12, unsigned char* owned, long long* carry_keys,
                             double* carry_vals);
int lbmdem_dist_table_substep(lbmdem_handle* h, const double* state12_full, const double* carry_vals,
                              const int* carry_has);
int lbmdem_vtk_place_owned(lbmdem_handle* h, float* fields11);
int lbmdem_write_vtk_fields(const char* dir, int nfile, int lx, int ly, const float* fields11);
int lbmdem_comm_write_vtk(lbmdem_handle* h, lbmdem_comm* c, const char* dir, int nfile);
/* Checkpoints of a strip decomposition: every rank saves its own handle (lbmdem_checkpoint_save: its strip, the grains
 * as it holds them, ownership masks, message capacities) and lbmdem_checkpoint_load brings it back with the grains
 * distributed again. The "previous contact" carries are agreed over the ranks first: lbmdem_dist_export_carries gives
 * a rank's youngest records and its standing values, lbmdem_dist_set_carries installs the result (per carry the
 * greatest key pair over all ranks, else rank 0's standing value); lbmdem_comm_sync_carries = both over RCCL. */
int lbmdem_dist_export_carries(lbmdem_handle* h, long long* carry_keys, double* carry_vals, double* carry_standing);
int lbmdem_dist_set_carries(lbmdem_handle* h, const double* carry3);
int lbmdem_comm_sync_carries(lbmdem_handle* h, lbmdem_comm* c);
/* bitwise merge (integer sum) of host buffers whose non-zero bits are disjoint across the ranks; in place */
int lbmdem_comm_allreduce_bits(lbmdem_comm* c, void* host_buf, size_t nbytes);
/* sum of host values over the ranks (check_density / final_density; not on the step path) */
int lbmdem_comm_allreduce_sum(lbmdem_comm* c, double* values, int n);
/* a grouped send + receive of `doubles` values from this rank to itself on a side stream while another stream is
 * busy (the transport exercised on a one-GPU box); with several ranks also the step's own pattern: on every lane one
 * grouped exchange with both neighbours, all lanes in flight at once, payload checked. Collective over the ranks. */
int lbmdem_comm_selftest(lbmdem_comm* c, int doubles);

/* Device pointer to the 3*n hydrodynamic-force table (fhf1[n], fhf2[n], fhf3[n]) and to the
 * n-entry ownership mask (1 = this rank computed the grain) for the cross-rank combine. */
int lbmdem_fhf_device(lbmdem_handle* h, void** fhf, void** owner_mask);
/* Cross-rank combine of the hydrodynamic forces: forces_fluid writes the forces of the grains this
 * rank owns and exact zeros for the others, so a bit-wise integer SUM all-reduce of the exported
 * table (3*n doubles viewed as int64) reconstructs every grain's force exactly. export/import copy
 * the table to/from a caller-provided DEVICE buffer on the handle's stream. */
int lbmdem_fhf_export(lbmdem_handle* h, void* dev_buf);
int lbmdem_fhf_import(lbmdem_handle* h, const void* dev_buf);

const char* lbmdem_last_error(void);
const char* lbmdem_version(void);

/* ---- test aids -----------------------------------------------------------------------------------
 * Nothing a host needs; the GPU test suite checks the library's short cuts against its own slow paths through these.
 * Collected here so that the list above is the product's surface:
 *   lbmdem_set_change_mask(h, 2)   every fused launch is repeated with both maps read everywhere and compared (above)
 *   lbmdem_set_dem_chain(h, -1)    runs of sub-steps without the rasterisation at their end (above)
 *   lbmdem_set_obst_update(h, 0), lbmdem_set_dem_chain(h, 0), lbmdem_set_change_mask(h, 0), lbmdem_set_force_mode(h, 1)
 *                                  the A/B switches of bench.py (--obst-update, --dem-chain, --change-mask, --force-mode)
 *   the *_stats / lbmdem_dem_chain_paints / lbmdem_dem_chain_recoveries counters
 *   LBMDEM_RCCL_LIBRARY            (environment) the RCCL library lbmdem_comm_create loads; the tests point it at
 *                                  tests/rccl_shim to run several ranks on ONE GPU; announced on stderr when set
 * and, only here: */
/* after every sub-step, grains this rank does not integrate are overwritten with NaN (strip tests: nothing may read them) */
int lbmdem_dist_set_poison(lbmdem_handle* h, int on);

#pragma pop_macro("scale")
#pragma pop_macro("ly")
#pragma pop_macro("lx")

#ifdef __cplusplus
}
#endif
#endif
