/*
 * pcp_hip.h -- C ABI of libpcp_hip.so: the MI355X (gfx950) kernels of the PointPillars collaborative-perception hot path.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless its name ends in _host;
 *   - the caller owns every buffer, including scratch ("workspace"), whose size it obtains from the matching
 *     *_workspace_bytes() query; nothing is allocated, freed or synchronised inside a call (hipGraph-capturable);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it;
 *   - the return value is a status (PCP_OK = 0); the library never calls exit() (the reference's op does:
 *     pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:14-38), never reads the environment (no source of it calls getenv; hipcub's radix sort
 *     inside the training entry pcp_hunter_losses brings rocPRIM's own ROCPRIM_USE_ATOMIC_BLOCK_ID query along), and is re-entrant: what a launch does is a function
 *     of its arguments and of the option table below (A/B and diagnostic overrides the host sets explicitly; all default to "built-in
 *     rule"), plus a read-only cache of each device's CU count keyed by device ordinal;
 *   - dense maps are NHWC float32 ("pixel-major": channels contiguous), the layout the MFMA implicit-GEMM tiles read
 *     coalesced; the Python host exposes them to callers as NCHW-shaped channels_last views.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference repo root).
 */
#ifndef PCP_HIP_H
#define PCP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCP_ABI_VERSION 1

enum {
  PCP_OK = 0,
  PCP_ERR_ARG = 1,          /* bad argument (null pointer, size, alignment, unsupported shape) */
  PCP_ERR_WORKSPACE = 2,    /* workspace too small */
  PCP_ERR_LAUNCH = 3,       /* hipGetLastError() after a launch was not hipSuccess */
  PCP_ERR_UNSUPPORTED = 4   /* valid request this build has no kernel for */
};

int pcp_abi_version(void);
const char *pcp_status_string(int status);

/* The only process-wide mutable state of the library: integer overrides of built-in launch rules, for A/B measurements and tests.
 * value < 0 restores the built-in rule (the initial state of every option); reads and writes are atomic, a launch reads an option once.
 * The Python host (pcp_amd/lib.py) maps the PCP_* environment variables of earlier rounds onto these ONCE, when it loads the library. */
enum {
  PCP_OPT_PFN_CROWD = 0,        /* pcp_pillarise_rows: records from which a pillar gets a workgroup of pcp_pfn_rows (0 = never; >= 64; built-in 192) */
  PCP_OPT_PFN_CROWD_BLOCKS = 1, /* pcp_pfn_rows: workgroups at the front of the grid that run the crowded pillars (built-in 128) */
  PCP_OPT_PFN_WPS = 2,          /* pcp_pfn_rows: waves per SIMD the launch is held to, 2 or 3 (built-in: by cloud size) */
  PCP_OPT_WINO4C_NW = 3,        /* pcp_conv3x3_winograd4c: 8 = the eight-wave 128-channel form where cout_pad % 128 == 0 (built-in: four waves) */
  PCP_OPT_MP_TH16_MIN = 4,      /* pcp_mp_conv3x3: work items from which the 16-row item is used (built-in 256) */
  PCP_OPT_MP_DIAG = 5,          /* pcp_mp_conv3x3: timing-only diagnostic builds of the kernel body (built-in 0) */
  PCP_OPT_VOX_AGGREGATE = 6,    /* pcp_voxelize / pcp_pillarise_rows: 1 = count a workgroup's rows in an LDS hash table first, one global atomic per distinct
                                   (workgroup, cell) pair (built-in); 0 = one global atomic per row (rounds 1 - 5) */
  PCP_OPT_COUNT = 7
};
int pcp_set_option(int32_t option, int64_t value);
int64_t pcp_get_option(int32_t option);          /* the override, or -1 while the built-in rule applies (also for an unknown option) */

/* ------------------------------------------------------------------------------------------------------------------
 * a1 / a4  dynamic pillarisation.
 * Replaces: pcdet/models/backbones_3d/vfe/dynamic_pillar_vfe.py:96-108 and :137-147
 *           (floor((xy - min) / voxel), x/y range mask, merged id b*nx*ny + cx*ny + cy, torch.unique(sorted,
 *           return_inverse, return_counts)) -- bit exact, without a sort: dense per-cell counts + exclusive scan.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  float min_x, min_y, min_z;      /* POINT_CLOUD_RANGE[0:3] as float32 */
  float voxel_x, voxel_y, voxel_z;/* VOXEL_SIZE as float32 */
  int32_t nx, ny;                 /* grid (nz must be 1) */
  int32_t batch_size;             /* frames sharing one call; batch index = (int) points[:,0] */
} pcp_grid_t;

/* counters written by pcp_voxelize at counters[0..3]: P (pillars), N' (kept points), reserved, reserved */
#define PCP_VOX_COUNTERS 4

size_t pcp_voxelize_workspace_bytes(const pcp_grid_t *grid, int64_t max_points);

/* points: (n, row_stride) float32 rows [b, x, y, z, ...].  Outputs (caller-allocated for n rows, only the first P / N'
 * entries are written): voxel_coords (P,4) int32 [b,0,y,x]; unq_inv (N',) int64; unq_cnt (P,) int32 (may be NULL);
 * counters (PCP_VOX_COUNTERS,) int32.  The workspace keeps the bucketed point order for pcp_pfn_scatter. */
int pcp_voxelize(const float *points, int64_t n, int32_t row_stride, const pcp_grid_t *grid,
                 void *workspace, size_t workspace_bytes,
                 int32_t *voxel_coords, int64_t *unq_inv, int32_t *unq_cnt, int32_t *counters, void *stream);
/* The passes of pcp_voxelize behind its first one, for rows whose cell ids and per-cell histogram are already in the workspace
 * (pcp_select_transform_compact with vox_grid).  No unq_inv in this mode. */
int pcp_voxelize_cells_ready(const float *points, int64_t n, int32_t row_stride, const pcp_grid_t *grid,
                             void *workspace, size_t workspace_bytes,
                             int32_t *voxel_coords, int32_t *unq_cnt, int32_t *counters, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a2 / a3 / a5  point feature build + two PFN layers + scatter to the dense BEV canvas, one kernel.
 * Replaces: dynamic_pillar_vfe.py:110-126 (scatter_mean, f_cluster, f_center, concat), :35-46 (PFNLayerV2 x2:
 *           Linear + BatchNorm1d(eval) + ReLU + torch_scatter.scatter_max) and
 *           pcdet/models/backbones_2d/map_to_bev/pointpillar_scatter.py:14-37 (canvas zero + indexed write).
 * BatchNorm is folded by the host: w0 (32, num_raw+6), b0 (32), w1 (64, 64), b1 (64), row-major float32.
 * canvas: (B, ny, nx, 64) NHWC, must be zero where no pillar lands (pcp_fill_zero or a previous pcp_canvas_clear);
 * pillar_features: (P, 64) or NULL.  Must follow pcp_voxelize on the same workspace and stream.
 * ------------------------------------------------------------------------------------------------------------------ */
/* A processing order for per-point kernels that gather from BEV maps: order[0..N') = the pillariser's bucket order (spatially
 * sorted), then the rows it masked; every row exactly once.  Must follow pcp_voxelize on the same workspace.  cursor_scratch: 1 int. */
/* Sorts the points of every pillar by row index inside the bucket order pcp_voxelize left in `workspace` (their slots come from an atomic
 * and differ from run to run; inference results do not depend on them, the training path's per-point GEMMs sum in this order): after it
 * the bucket order is a function of the input alone.  Must follow pcp_voxelize on the same workspace and stream. */
int pcp_voxelize_sort_pillar_rows(const pcp_grid_t *grid, void *workspace, int64_t n, void *stream);
int pcp_voxelize_row_order(const pcp_grid_t *grid, const void *workspace, int64_t n, int32_t *order, int32_t *cursor_scratch,
                           void *stream);

int pcp_pfn_scatter(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, const pcp_grid_t *grid,
                    const void *workspace, const float *w0, const float *b0, const float *w1, const float *b1,
                    float *pillar_features, float *canvas, void *stream);

/* Round 5: the same two steps rebuilt for the memory system (csrc/voxelize.hip, csrc/pfn_rows.hip).
 * pcp_pillarise_rows = pcp_voxelize (same outputs, voxel_coords / unq_inv / unq_cnt may each be NULL) + the kept rows written IN PILLAR
 * ORDER into the workspace (32- or 64-byte records: raw columns 1 .. num_raw of the row, pillar rank, cell, frame) + one 8-byte
 * descriptor per wave tile of pcp_pfn_rows.  Workspace: pcp_pillarise_rows_workspace_bytes (its front part is laid out as
 * pcp_voxelize lays it out: pcp_sparse_conv3x3_s2 and the training kernels read it unchanged).  flags: PCP_ROWS_CELLS_READY = the
 * rows' cell ids and histogram are already in the workspace (pcp_select_transform_compact with vox_grid; no unq_inv then),
 * PCP_ROWS_BUCKET_ORDER = also leave the bucket order (row indices grouped by pillar, pillars ascending) for pcp_voxelize_row_order; without
 * it the records of single-point pillars lie behind those of all multi-point pillars (counters[2], [3] = records of multi-point pillars,
 * single-point pillars; pcp_pfn_rows runs the singles through a path without per-pillar reductions).
 * pcp_pfn_rows: the fused feature build + PFN x2 + scatter of pcp_pfn_scatter on those records; one WAVE per ~30-point run of pillars,
 * no workgroup barriers, both layers on fp32 MFMA with the weights as the A operand.  canvas (B, ny, nx, 64): written COMPLETELY --
 * pillar rows and zero rows for the empty cells -- so it needs no zero fill and no pcp_canvas_clear; pillar_features (P, 64) or NULL.
 * num_raw in {3, 4, 5, 11}; nx, ny <= 65535.  Must follow pcp_pillarise_rows on the same workspace and stream.
 * Pillars of at least 192 records (option PCP_OPT_PFN_CROWD overrides, >= 64; 0 = never) are listed in the workspace by
 * pcp_pillarise_rows, their records tagged (sign bit of the rank field): pcp_pfn_rows runs those on the first 128 workgroups of its grid
 * (option PCP_OPT_PFN_CROWD_BLOCKS overrides), a workgroup per pillar
 * (bit-identical results; a LiDAR-like cloud has cells with hundreds of points next to the sensor). */
#define PCP_ROWS_CELLS_READY 1
#define PCP_ROWS_BUCKET_ORDER 2
size_t pcp_pillarise_rows_workspace_bytes(const pcp_grid_t *grid, int64_t max_points, int32_t num_raw);
int pcp_pillarise_rows(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, const pcp_grid_t *grid,
                       void *workspace, size_t workspace_bytes, int32_t *voxel_coords, int64_t *unq_inv, int32_t *unq_cnt,
                       int32_t *counters, int32_t flags, void *stream);
int pcp_pfn_rows(const pcp_grid_t *grid, const void *workspace, int64_t n, int32_t num_raw, const float *w0, const float *b0,
                 const float *w1, const float *b1, float *pillar_features, float *canvas, void *stream);
/* The reference's index tensors ON DEMAND from the workspace a pcp_pillarise_rows (num_raw as in that call, any flags, outputs NULL or not) or a
 * pcp_voxelize (num_raw = 0) with the same grid and n left behind -- the tables pcp_pfn_rows / pcp_sparse_conv3x3_s2 read, so what comes out is
 * the pillar list the maps were computed from (dynamic_pillar_vfe.py:104-108, 137-147).  Every output may be NULL:
 *   voxel_coords (n, 4) int32, first P rows written: [b, 0, y, x];  row_rank (n,) int32: pillar rank of input row i, -1 for a masked row
 *   (unq_inv = row_rank[row_rank >= 0]);  slot_rank / slot_canvas_row (n,) int32, first N' written (num_raw > 0 only): pillar rank and canvas row
 *   (b * ny + y) * nx + x carried by the record in slot s of the pillar-ordered rows;  counters (PCP_VOX_COUNTERS,) int32 as the workspace holds
 *   them: P, N', records of multi-point pillars, single-point pillars. */
int pcp_pillar_index_export(const pcp_grid_t *grid, const void *workspace, int64_t n, int32_t num_raw, int32_t *voxel_coords,
                            int32_t *row_rank, int32_t *slot_rank, int32_t *slot_canvas_row, int32_t *counters, void *stream);

/* The PillarFeatureNet variants none of the reference's configs use (WITH_DISTANCE, USE_ABSLOTE_XYZ False, NUM_FILTERS other than
 * [64, 64], any raw width) run as separate steps: pcp_voxelize, then
 * pcp_pfn_features: the feature rows of dynamic_pillar_vfe.py:110-126 in BUCKET ORDER, `fw` floats per row (zero padded), composition by
 *   flags: [points[:, 1:] or points[:, 4:], f_cluster, f_center, |xyz| if PCP_PFN_WITH_DISTANCE]; slot_pillar[s] = pillar rank of slot s;
 * then per PFNLayerV2 (:35-46) pcp_pointwise (Linear + folded BatchNorm + ReLU), pcp_segment_max (torch_scatter.scatter_max) and, for all
 * but the last layer, pcp_pfn_cat_pillar_max: out[r] = [y[r, :c], pillar_max[slot_pillar[r], :c], 0 ...]. */
#define PCP_PFN_ABSOLUTE_XYZ 1u
#define PCP_PFN_WITH_DISTANCE 2u
int pcp_pfn_features(const float *points, int64_t n, int32_t row_stride, int32_t num_raw, uint32_t flags, const pcp_grid_t *grid,
                     const void *vox_workspace, int32_t fw, float *fbuf, int32_t *slot_pillar, void *stream);
int pcp_pfn_cat_pillar_max(const float *y, int32_t ld_y, const float *pillar_max, int32_t ld_max, const int32_t *slot_pillar, int64_t rows,
                           int32_t c, float *out, int32_t ld_out, void *stream);

/* zero the canvas rows written by an earlier pcp_pfn_scatter (reads the pillar list still held in that call's
 * workspace; n = the point count of that call): P * 256 B instead of re-zeroing B * ny * nx * 256 B */
int pcp_canvas_clear(const pcp_grid_t *grid, const void *workspace, int64_t n, float *canvas, void *stream);
int pcp_fill_zero(void *ptr, size_t bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a6 / a7 / a12 / a14  dense 2D convolutions as fp32 MFMA implicit GEMM (v_mfma_f32_32x32x2_f32), BN folded,
 * bias + optional ReLU fused.
 * Replaces: cuDNN Conv2d/ConvTranspose2d + BatchNorm2d + ReLU stacks of
 *           pcdet/models/backbones_2d/base_bev_backbone.py:30-69, pcdet/models/dense_heads/center_head.py:24-29,75-82,
 *           pcdet/models/bev_layers/v2x_fusion_disco.py:51-63, pcdet/models/bev_layers/hunter_jr.py:132,149-152.
 * Tensors are NHWC; `ld*` = floats between consecutive pixels (>= channels) so channel slices of a wider buffer
 * (the 384-channel concat, the 768-channel HunterJr input) are addressed in place.
 * Packed weights (host packs once): 3x3 -> [Cin/16][9][cout_pad][16]; pointwise -> [K/16][n_pad][16]; bias [cout_pad].
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch, in_h, in_w;      /* input spatial size */
  int32_t cin, cout;              /* cin % 16 == 0 */
  int32_t cout_pad;               /* padded output channels of the packed weights (multiple of 32) */
  int32_t stride;                 /* 1 or 2 (3x3, padding 1) */
  int32_t ld_in, ld_out;          /* pixel strides in floats */
  int32_t relu;                   /* 1: max(x, 0) after bias */
} pcp_conv3x3_t;

int pcp_conv3x3(const pcp_conv3x3_t *desc, const float *in, const float *w_packed, const float *bias, float *out,
                void *stream);

/* a5 + a6, first backbone layer from the pillar list: ZeroPad2d(1) + Conv2d(64, cout, 3, stride 2) + folded BN + ReLU
 * (base_bev_backbone.py:36-44) WITHOUT the dense canvas of pointpillar_scatter.py:14-37 -- the occupied input cells of every tap are
 * found in the cell -> pillar-rank table pcp_voxelize left in its workspace, their pillar rows gathered and multiplied on the matrix
 * cores, products added per output pixel in a fixed order (deterministic, no atomics).
 * pillar_features: (P, 64) in pillar-rank order (pcp_pfn_scatter's optional output); grid / vox_workspace / n: as passed to pcp_voxelize;
 * w_packed: [9 (ky*3+kx)][64][64 (cin)] float32 (rows >= cout zero), bias [64]; cout <= 64, cout % 4 == 0;
 * out: (B, (ny-1)/2+1, (nx-1)/2+1, ld_out) NHWC. */
int pcp_sparse_conv3x3_s2(const float *pillar_features, const pcp_grid_t *grid, const void *vox_workspace, int64_t n,
                          const float *w_packed, const float *bias, int32_t cout, int32_t relu, float *out, int32_t ld_out,
                          void *stream);

/* Same operation for stride 1 as a fused Winograd F(2x2,3x3) kernel (2.25x fewer multiplies, still fp32 MFMA with fp32
 * accumulation; transforms add ~1e-6 relative rounding).  Needs cin % 8 == 0, cout_pad % 64 == 0 and weights packed as
 * U = G g G^T: [cin/8][16 (i*4+j)][cout_pad][8]. */
int pcp_conv3x3_winograd(const pcp_conv3x3_t *desc, const float *in, const float *u_packed, const float *bias, float *out,
                         void *stream);

/* Same operation, same arithmetic (fused Winograd F(2x2,3x3), fp32 MFMA), "wave-stationary" decomposition for the short-K layers
 * (base_bev_backbone.py:30-69 blocks at 64 / 128 channels, center_head.py:24-29 branches, v2x_fusion_disco.py:51-58 compressor): a wave
 * keeps 32 tiles x 32 output channels x 8 Winograd positions in registers for the whole contraction and transforms its own A operand in
 * registers (no V round trip through LDS, no per-slice barrier).  cin % 32 == 0, cout_pad % 64 == 0, ld_in % 4 == 0, `in` and `u_packed`
 * 16-byte aligned; weights packed by pcp_amd/pack.py::pack_conv3x3_winograd_ws as U = G g G^T:
 * [cin/2][cout_pad/32][2 (position rows {0,1} | {2,3})][2 (row)][64 (channel parity, cout)][4 (position column)].
 * _supported: 1 if `desc` meets those shape rules; _plan: the kernel variant (tile groups per workgroup: 1 = 8x16 px x 128 ch,
 * 2 = 16x16 px x 64 ch) and the flops the launch executes on the matrix pipe (padding included); either output may be NULL. */
int pcp_conv3x3_winograd_ws_supported(const pcp_conv3x3_t *desc);
int pcp_conv3x3_winograd_ws(const pcp_conv3x3_t *desc, const float *in, const float *u_packed, const float *bias, float *out,
                            void *stream);
int pcp_conv3x3_winograd_ws_plan(const pcp_conv3x3_t *desc, int32_t *variant, double *executed_flops);

/* Same operation as ONE fused Winograd F(4x4,3x3) launch (csrc/wino4f.hip): 36 products per 4x4 output tile (1.78x fewer matrix flops
 * than F(2x2), 4x fewer than the direct form), fp32 MFMA with fp32 accumulation, transform rounding ~1e-5 of the output scale; nothing
 * goes through HBM between the transforms and the products (no workspace).  Built for the 64 / 128-channel layers of
 * base_bev_backbone.py:30-69, center_head.py:24-29,75-82 and v2x_fusion_disco.py:51-58.  cin % 8 == 0, cout % 4 == 0, cout_pad % 64 == 0,
 * ld_in % 4 == 0, ld_out % 4 == 0, `in`, `out`, `bias` and `u_packed` 16-byte aligned; weights packed by pcp_amd/pack.py::pack_conv3x3_winograd4f as U = G g G^T:
 * [cin/8][36 (i*6+j)][cout_pad][8].  _plan: the flops the launch executes on the matrix pipe (padding tiles / channels included). */
int pcp_conv3x3_winograd4f(const pcp_conv3x3_t *desc, const float *in, const float *u_packed, const float *bias, float *out,
                           void *stream);
int pcp_conv3x3_winograd4f_plan(const pcp_conv3x3_t *desc, double *executed_flops);

/* The same fused F(4x4,3x3) convolution with TWO four-wave workgroups per CU (csrc/wino4h.hip): an item is 16 x 16 output pixels x 64
 * channels on v_mfma_f32_16x16x4_f32, so one workgroup's prologue / epilogue runs under the other one's MFMAs.  Same descriptor rules as
 * pcp_conv3x3_winograd4f; weights packed by pcp_amd/pack.py::pack_conv3x3_winograd4h: [cin/8][36 (i*6+j)][cout_pad/64][64 lanes][8], lane
 * l = 16 kq + c, value 2 nb + ks = U[i][j][input channel 8 s + 4 ks + kq][output channel 64 n + 16 nb + c]. */
int pcp_conv3x3_winograd4h(const pcp_conv3x3_t *desc, const float *in, const float *u_packed, const float *bias, float *out,
                           void *stream);
int pcp_conv3x3_winograd4h_plan(const pcp_conv3x3_t *desc, double *executed_flops);

/* The same items and workgroups with the four waves split over OUTPUT CHANNELS instead of Winograd positions (csrc/wino4c.hip): a lane holds
 * all 36 position values of four consecutive channels of one tile, so the output transform, bias and ReLU run in registers -- no LDS image
 * of the accumulators, no epilogue barriers.  Bit-identical outputs to pcp_conv3x3_winograd4h.  Same descriptor rules; weights packed by
 * pcp_amd/pack.py::pack_conv3x3_winograd4c: [cin/8][cout_pad/16][18 position pairs][64 lanes][4], lane l = 16 kq + c, value 2 e + ks =
 * U[position 2 q + e][input channel 8 s + 4 ks + kq][output channel 16 g + c]. */
int pcp_conv3x3_winograd4c(const pcp_conv3x3_t *desc, const float *in, const float *u_packed, const float *bias, float *out,
                           void *stream);
int pcp_conv3x3_winograd4c_plan(const pcp_conv3x3_t *desc, double *executed_flops);

/* Measurement query (bench.py's roofline line): which instantiation pcp_conv3x3_winograd launches for `desc` (variant = 1: 32-tile
 * workgroups k_conv3x3_wino<1>, 2: 64-tile workgroups k_conv3x3_wino<2>) and the flops that launch EXECUTES on the matrix pipe
 * (16 products per 2x2 output tile and (cin, cout) pair, padding tiles and channels included).  Either output may be NULL. */
int pcp_conv3x3_winograd_plan(const pcp_conv3x3_t *desc, int32_t *variant, double *executed_flops);

/* Same operation for the wide stride-1 layers (cin >= 256) as Winograd F(4x4,3x3) in three launches through a caller-owned
 * workspace: input transform -> 36 batched fp32-MFMA GEMMs [tiles x cin] x [cin x cout] -> output transform + bias + ReLU.  4x fewer
 * multiplies than the direct form, fp32 arithmetic and accumulation throughout; transform rounding ~2e-5 of the output scale.
 * cin % 32 == 0, cout % 4 == 0, cout_pad % 128 == 0, ld_in % 4 == 0, ld_out % 4 == 0, all pointers 16-byte aligned;
 * weights packed by pcp_amd/pack.py::pack_conv3x3_winograd4 as U = G g G^T: [36 (i*6+j)][cout_pad][cin].
 * workspace: pcp_conv3x3_winograd4_workspace_bytes(desc) bytes of device memory (contents undefined before and after). */
int pcp_conv3x3_winograd4_workspace_bytes(const pcp_conv3x3_t *desc, size_t *bytes);
int pcp_conv3x3_winograd4(const pcp_conv3x3_t *desc, const float *in, const float *u_packed, const float *bias, float *out,
                          void *workspace, void *stream);
/* Measurement variant (bench.py's instrumented pass): same launches bracketed by HIP events on `stream`; waits for them and returns
 * the duration of each launch in milliseconds (stage_ms_host[3] = input transform, batched GEMM, output transform) and the flops the
 * GEMM launch executes (2 * 36 * tiles_padded * cin * cout_pad; may be NULL). */
int pcp_conv3x3_winograd4_timed(const pcp_conv3x3_t *desc, const float *in, const float *u_packed, const float *bias, float *out,
                                void *workspace, void *stream, float *stage_ms_host, double *gemm_flops_host);

/* OPT-IN alternative arithmetic for the same operation (stride 1 or 2): fp32 tensors in and out, products on the BF16 matrix cores with
 * every operand split in two bf16 halves (hi + lo, 16 mantissa bits) and three MFMAs per product, fp32 accumulation: ~1e-5 relative
 * error (the reference's own GPU path, cuDNN with TF32 allowed, keeps 10 bits).  cin % 16 == 0, cout_pad % 64 == 0; weights packed by
 * pcp_amd/pack.py::pack_conv3x3_bf16x3 as bf16 [cin/16][cout_pad/64][hi|lo][9][2][64][8].  Never selected by default. */
int pcp_conv3x3_bf16x3(const pcp_conv3x3_t *desc, const float *in, const void *w_packed, const float *bias, float *out, void *stream);
/* Same kernel with PLAIN bf16 products (only the hi halves: 8 mantissa bits, ~3e-3 relative error, fp32 accumulation; same packed
 * weights): the mixed-precision TRAINING arithmetic BASELINE.json names for config 5 ("training loop bf16").  Opt-in through
 * PCP_CONV_ALGO=bf16 on the training layers' forward / data-gradient convolutions; never used where parity with the reference's
 * fp32 results is claimed. */
int pcp_conv3x3_bf16(const pcp_conv3x3_t *desc, const float *in, const void *w_packed, const float *bias, float *out, void *stream);

/* Grouped 3x3 conv with tiny outputs (the final convs of the CenterHead branches, center_head.py:39; HunterJr's 768 -> 2
 * weight conv, hunter_jr.py:151): group g reads input channels [g cin_g, (g+1) cin_g) and produces output channels
 * [off[g], off[g+1]) (1..4 each), bias added, no activation.  cin_per_group % 64 == 0.
 * weights: [n_out][9 (ky*3+kx)][cin_per_group] float32; group_out_offsets_host: (groups + 1) int32 on the HOST; groups <= 8. */
int pcp_conv3x3_grouped_small(const float *in, int32_t batch, int32_t h, int32_t w, int32_t ld_in, int32_t groups,
                              int32_t cin_per_group, const int32_t *group_out_offsets_host, const float *weights, const float *bias, float *out,
                              int32_t ld_out, void *stream);

enum {
  PCP_PW_PLAIN = 0,       /* rows = pixels (or points): out[m, n] = sum_k in[m, k] w[n, k]                       */
  PCP_PW_SPACE2DEPTH = 1, /* Conv2d k=2 s=2: K = 4*cin gathered from the 2x2 input block of each output pixel    */
  PCP_PW_DEPTH2SPACE = 2  /* ConvTranspose2d k=2 s=2: N = 4*cout scattered to the 2x2 output block               */
};

typedef struct {
  int32_t mode;
  int64_t rows;                   /* PLAIN: number of rows; otherwise ignored (derived from batch/h/w) */
  int32_t batch, in_h, in_w;      /* spatial modes: input size */
  int32_t cin, cout, cout_pad;    /* per-tap channels; cout_pad = padded n of the packed weights per tap */
  int32_t ld_in, ld_out;
  int32_t relu;
  /* PLAIN only (NULL / 0 otherwise): a second K source -- channels [k_split, cin) are read from in2 at offset
   * k - k_split (lets cat([a, b], channel) feed a 1x1 conv without materialising the cat) -- and a residual row
   * added after bias / activation (out = act(x W^T + b) + residual). */
  const float *in2;
  int32_t ld_in2, k_split;
  const float *residual;
  int32_t ld_res;
} pcp_pointwise_t;

int pcp_pointwise(const pcp_pointwise_t *desc, const float *in, const float *w_packed, const float *bias, float *out,
                  void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a8  CenterHead decode: sigmoid / exp / atan2, top-K over the heat map, box assembly, range + score mask.
 * Replaces: pcdet/models/dense_heads/center_head.py:302-333 and pcdet/models/model_utils/centernet_utils.py:127-214.
 * head: (B, H, W, ld) NHWC holding center(2), center_z(1), dim(3), rot(2) = (cos, sin) and hm(num_class) at the
 * channel offsets given in the descriptor.  Ties in the top-K go to the lower flat (class, cell) index.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch, h, w, ld;        /* head buffer (B, H, W, ld) NHWC */
  int32_t num_class;              /* heat-map channels; num_class * h * w <= 16384 */
  int32_t ch_center, ch_z, ch_dim, ch_rot, ch_hm;   /* channel offsets inside a pixel */
  int32_t k;                      /* MAX_OBJ_PER_SAMPLE (<= 1024) */
  float stride;                   /* FEATURE_MAP_STRIDE */
  float voxel_x, voxel_y, min_x, min_y;
  float limit[6];                 /* POST_CENTER_LIMIT_RANGE (inclusive both ends) */
  int32_t use_score_thresh;       /* SCORE_THRESH is not None */
  float score_thresh;             /* strict > */
  int32_t activated;              /* 0: hm holds logits and dim log-sizes (the head's raw maps); 1: hm holds sigmoid scores and dim exp'ed
                                   * sizes -- the call convention of centernet_utils.decode_bbox_from_heatmap (center_head.py:312-333) */
} pcp_decode_t;

size_t pcp_decode_workspace_bytes(const pcp_decode_t *desc);
/* Outputs per frame b (capacity k rows each): boxes (B,k,7), scores (B,k), labels (B,k) int32 0-based class (may be
 * NULL), cell (B,k) int32 flat y*w+x (may be NULL), count (B,) int32 -- rows in descending score order. */
int pcp_centerhead_decode(const pcp_decode_t *desc, const float *head, void *workspace, size_t workspace_bytes,
                          float *boxes, float *scores, int32_t *labels, int32_t *cell, int32_t *count, void *stream);

/* a8/a9 tail: final per-frame detections.  Replaces the per-frame python of center_head.py:335-357 (boxes[keep], scores[keep],
 * class_id_mapping[labels[keep]] + 1, torch.cat over heads) with one launch for all frames and heads.  Head h contributes its
 * min(keep_count[b], keep_max) kept candidates in keep order; heads are concatenated in array order.
 * out_boxes (B, out_max, 7), out_scores (B, out_max), out_labels (B, out_max) int64 1-based, out_count (B,) int32;
 * rows beyond out_count[b] are left untouched. */
#define PCP_DET_MAX_HEADS 8
typedef struct {
  const float *boxes;         /* (B, k, 7)  decode output */
  const float *scores;        /* (B, k) */
  const int32_t *labels;      /* (B, k) 0-based class inside the head; NULL = 0 */
  const int32_t *keep;        /* (B, keep_max) indices into the k candidates (pcp_nms_rotated) */
  const int32_t *keep_count;  /* (B,) */
  const int32_t *class_map;   /* head class -> global 0-based class id (device); NULL = identity */
  int32_t k, keep_max;
} pcp_det_head_t;
int pcp_gather_detections(const pcp_det_head_t *heads, int32_t n_heads, int32_t batch, int32_t out_max, float *out_boxes,
                          float *out_scores, int64_t *out_labels, int32_t *out_count, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a9  rotated BEV IoU + NMS, entirely on the device.
 * Replaces: pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-17 -> iou3d_nms.cpp:52-136 + iou3d_nms_kernel.cu:236-311
 *           (boxes_overlap_bev_gpu, boxes_iou_bev_gpu, nms_gpu) and the score sort of iou3d_nms_utils.py:92-95.
 * pcp_nms_rotated: `batch` independent frames; boxes (batch, n_max, 7) + scores (batch, n_max); frame f uses its first
 *   n_dev[f] rows when n_dev != NULL (else n_max).  Per frame: sort by descending score (bitonic in LDS, ties by lower
 *   index; skipped when scores == NULL = "already sorted"), cut to pre_max, build the suppression bit mask (one
 *   wavefront per 64-bit word), run the greedy sweep in one wavefront.  keep (batch, post_max) int32 indices into the
 *   frame's INPUT order; keep_count (batch,) int32.  n_max <= 4096.
 * ------------------------------------------------------------------------------------------------------------------ */
size_t pcp_nms_workspace_bytes(int32_t n_max, int32_t batch);
int pcp_nms_rotated(const float *boxes, const float *scores, int32_t batch, int32_t n_max, const int32_t *n_dev,
                    float thresh, int32_t pre_max, int32_t post_max, void *workspace, size_t workspace_bytes,
                    int32_t *keep, int32_t *keep_count, void *stream);
/* The same pipeline with the AXIS-ALIGNED IoU of nms_normal_gpu (iou3d_nms_api.cpp:16 -> iou3d_nms.cpp:139-188 +
 * iou3d_nms_kernel.cu:314-372: x / y extents only, the heading is ignored). */
int pcp_nms_normal(const float *boxes, const float *scores, int32_t batch, int32_t n_max, const int32_t *n_dev,
                   float thresh, int32_t pre_max, int32_t post_max, void *workspace, size_t workspace_bytes,
                   int32_t *keep, int32_t *keep_count, void *stream);
/* mode 0: overlap area (boxes_overlap_bev_gpu), mode 1: IoU (boxes_iou_bev_gpu); out (na, nb) float32 */
int pcp_boxes_bev_pairwise(const float *a, int32_t na, const float *b, int32_t nb, int32_t mode, float *out,
                           void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a11 / a12  DiscoNet mid fusion: nearest affine warp and pixel-wise softmax-weighted sum.
 * Replaces: pcdet/models/bev_layers/v2x_fusion_disco.py:29-45 (transform_bev_img = affine_grid + grid_sample nearest,
 *           align_corners=False, zero padding) and :104-115 (softmax over agents + weighted sum).
 * theta: (2,3) float32 row-major (host computes it exactly as :32-35).  src/dst: (H, W, ld) NHWC, c channels copied.
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_warp_nearest(const float *src, float *dst, int32_t h, int32_t w, int32_t c, int32_t ld_src, int32_t ld_dst,
                     const float *theta_host, int32_t accumulate, void *stream);
/* The same warp for n_jobs (source map, destination map, theta) triples of one geometry in ONE launch (the (agent, frame) pairs of a DiscoNet
 * forward, v2x_fusion_disco.py:88-101): src_host / dst_host are HOST arrays of device pointers, theta_host n_jobs x 6 floats. */
int pcp_warp_nearest_batch(const float *const *src_host, float *const *dst_host, const float *theta_host, int32_t n_jobs, int32_t h,
                           int32_t w, int32_t c, int32_t ld_src, int32_t ld_dst, int32_t accumulate, void *stream);
/* Round 6: the same launch with the affines in DEVICE memory (theta_dev: n_jobs x 6 floats) -- inside a captured hipGraph the poses are then
 * data, not kernel arguments: the host rewrites the table before a replay and one capture serves every pose set. */
int pcp_warp_nearest_batch_dev(const float *const *src_host, float *const *dst_host, const float *theta_dev, int32_t n_jobs, int32_t h,
                               int32_t w, int32_t c, int32_t ld_src, int32_t ld_dst, int32_t accumulate, void *stream);
/* maps_host: HOST array of n_agents (<= 16) DEVICE pointers, each (pixels, ld_map); weights: (pixels, ld_w) logits, one
 * column per agent; out[p, :] = sum_a softmax_a(weights[p, :])[a] * maps[a][p, :] */
int pcp_softmax_fuse(const float *const *maps_host, int32_t n_agents, const float *weights, int32_t ld_w, int64_t pixels,
                     int32_t c, int32_t ld_map, int32_t ld_out, float *out, void *stream);
/* a12 (round 3): pixel weightor + softmax over the maps + weighted sum in ONE launch (v2x_fusion_disco.py:8-26,85,104,107-115).
 * maps_host: host array of n_maps device pointers (map 0 = the compressed ego map, the others the warped agent maps), each (pixels, ld_map)
 * with c = 128 channels.  For every map a: l_a = relu(w3 . relu(w2 . relu(w1 . [map_0 | map_a] + b1) + b2) + b3) with BatchNorm folded
 * by the host (w1 (64, 2c) row-major, w2 (16, 64), w3 (16), b3 (1)); out[p] = sum_a softmax_a(l)[p] * map_a[p]  (pixels, ld_out).
 * logits (optional, (pixels, ld_w)): the pre-softmax weights l_a.  n_maps <= 16; c != 128 -> PCP_ERR_UNSUPPORTED. */
int pcp_disco_weight_fuse(const float *const *maps_host, int32_t n_maps, int32_t c, int32_t ld_map, int64_t pixels,
                          const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                          float *out, int32_t ld_out, float *logits, int32_t ld_w, void *stream);
/* The same launch with maps that may not exist, decided on the device (DiscoNet under hipGraph): map a leaves the softmax -- weight exactly
 * 0, the others exactly the softmax over the maps that exist -- when live_index_host[a] >= 0 and live[live_index_host[a]] == 0 (`live`: the
 * flags of pcp_agent_frame_live; the ego map passes -1).  What the reference does by not having the agent in batch_dict['bev_img'] at all
 * (v2x_fusion_disco.py:83). */
int pcp_disco_weight_fuse_live(const float *const *maps_host, int32_t n_maps, int32_t c, int32_t ld_map, int64_t pixels,
                               const float *w1, const float *b1, const float *w2, const float *b2, const float *w3, const float *b3,
                               float *out, int32_t ld_out, const int32_t *live_index_host, const int32_t *live, void *stream);


/* ------------------------------------------------------------------------------------------------------------------
 * a14  HunterJr point <-> BEV ops.
 * Replaces: pcdet/models/bev_layers/hunter_toolbox.py:8-39,94-127 (bilinear sampling of a BEV map at point locations)
 *           and :65-91 (bev_scatter: strict float mask, truncation, torch.unique + scatter_mean).
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_bev_sample_bilinear(const float *bev, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_bev,
                            const float *points, int64_t n, int32_t row_stride, float min_x, float min_y, float pix_x,
                            float pix_y, const uint8_t *row_mask, float *out, int32_t ld_out, void *stream);
/* Fused HunterJr point head (hunter_jr.py:78-101 on top of the sampling above): pf = bilinear(bev, points) (n, ld_pf) is
 * written once; h = relu(pf W1^T + b1); f = relu(h W2^T + b2) + pf; head = f Wh^T + bh (n, n_out) = [cls(3) | flow(3) | embed(2)].
 * Weights are BN-folded row-major float32: w1 (hidden, c), w2 (c, hidden), wh (n_out, c).  This build: c = 384, hidden = 32,
 * n_out = 8 (PCP_ERR_UNSUPPORTED otherwise; the unfused ops remain available). */
int pcp_hunter_point_head(const float *bev, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_bev,
                          const float *points, int64_t n, int32_t row_stride, float min_x, float min_y, float pix_x,
                          float pix_y, const float *w1, const float *b1, const float *w2, const float *b2, const float *wh,
                          const float *bh, int32_t hidden, int32_t n_out, float *pf, int32_t ld_pf, float *head,
                          void *stream);
/* Extended form.  order (NULL: index order): visit the rows order[0 .. *order_count) (order_count NULL: n; e.g.
 * pcp_voxelize_row_order) -- results are written at the ORIGINAL row index, only the gather locality changes.
 * apply_flow != 0 fuses pcp_hunter_apply_flow and the masked re-sampling that follows it in correct_bev_image (hunter_jr.py:257-275):
 * rows predicted dynamic foreground get points[:, 1:4] += flow IN PLACE, dyn_mask[i] (may be NULL) = 1, and their pf row is re-sampled
 * at the corrected location (rows of foreign frames are moved but not re-sampled, like the unfused pair). */
int pcp_hunter_point_head_ex(const float *bev, int32_t batch, int32_t h, int32_t w, int32_t c, int32_t ld_bev, float *points, int64_t n,
                             int32_t row_stride, float min_x, float min_y, float pix_x, float pix_y, const float *w1, const float *b1,
                             const float *w2, const float *b2, const float *wh, const float *bh, int32_t hidden, int32_t n_out, float *pf,
                             int32_t ld_pf, float *head, const int32_t *order, const int32_t *order_count, int32_t apply_flow,
                             float flow_thresh, uint8_t *dyn_mask, void *stream);
size_t pcp_bev_scatter_mean_workspace_bytes(int32_t batch, int32_t h, int32_t w, int64_t n);
int pcp_bev_scatter_mean(const float *points, int64_t n, int32_t row_stride, const float *feat, int32_t ld_feat,
                         int32_t c, int32_t batch, int32_t h, int32_t w, float min_x, float min_y, float pix_x,
                         float pix_y, void *workspace, size_t workspace_bytes, float *out, int32_t ld_out, void *stream);
/* hunter_jr.py:257-265: p = sigmoid(head[:, 0:3]); dynamic foreground = max p > thresh and argmax == 2; for those rows
 * points[:, 1:4] += head[:, 3:6] IN PLACE; row_mask[i] = 1/0.  head: (n, ld_head) = [cls(3), flow(3), ...]. */
int pcp_hunter_apply_flow(float *points, int64_t n, int32_t row_stride, const float *head, int32_t ld_head, float thresh,
                          uint8_t *row_mask, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * a13  BEVMaker input preparation: rows of `agent` (last column == agent) are copied with xyz mapped by the frame's
 * 3x4 pose (x' = R x + t, row-major R|t per batch element, float32); every other row gets batch index -1 so that
 * pcp_voxelize masks it.  Replaces pcdet/models/bev_layers/bev_maker.py:168-179 (boolean-mask copy + per-frame matmul).
 * poses_host: (batch, 12) float32 on the HOST; present_host: (batch,) uint8, 0 = agent absent from that frame.
 * out_batch_offset is added to the frame index of the rows kept: several agents' clouds can then be stacked into ONE pass of the shared
 * frozen chain (agent slot i -> frames [i * batch, (i + 1) * batch)), which is bit-identical per frame to separate passes.
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_select_transform_points(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, float agent,
                                int32_t batch, const float *poses_host, const uint8_t *present_host, float *out,
                                int32_t out_batch_offset, void *stream);
/* Which agents have points (bev_maker.py:156 `torch.unique(points[:, -1])`, a sort + host sync in the reference): out2[0] = bit mask
 * of the integer values 0..63 found in column `col`, out2[1] = number of rows holding anything else (then the caller sorts).
 * out2: 2 x uint64 on the device, 16-byte aligned; zeroed by the call. */
int pcp_column_id_mask(const float *points, int64_t n, int32_t row_stride, int32_t col, uint64_t *out2, void *stream);
/* The same with the number of rows per id: out66[0] = presence mask, out66[1] = rows outside 0..63, out66[2 + i] = rows whose value
 * truncates to i.  One read-back then sizes every per-agent selection of a forward on the host (bev_maker.py:156,168-169).
 * out66: 66 x uint64 on the device, 16-byte aligned; zeroed by the call. */
int pcp_column_id_counts(const float *points, int64_t n, int32_t row_stride, int32_t col, uint64_t *out66, void *stream);

/* a13 as a STABLE COMPACTION (round 3; replaces bev_maker.py:168-190 for all agents of one stacked pass): for slots s = 0 .. n_slots-1
 * the rows of agent agents_host[s] whose frame b has present_host[s * batch + b] != 0 are written to `out` in input order, slot s
 * directly behind slot s-1 (= torch.cat([points[mask_s] for s])), with frame index b + s * batch and xyz mapped by the 3x4 pose
 * poses_host[(s * batch + b) * 12 ..] (same rounding order as pcp_select_transform_points).  Rows [total, out_rows) of `out` get frame
 * index -1 (pcp_voxelize drops them); slot_start (n_slots + 1 int32 on the device, may be NULL) receives the first row of every slot
 * and the total.  n_slots <= 8, n_slots * batch <= 64 (the pose table is a kernel argument).  workspace: device scratch of
 * pcp_select_transform_compact_workspace_bytes(n, n_slots).
 * vox_grid != NULL (batch_size = n_slots * batch): the cell id of every written row and the per-cell histogram are emitted into
 * vox_workspace, laid out as pcp_voxelize lays it out for (vox_grid, out_rows); follow with pcp_voxelize_cells_ready on the same
 * arguments instead of pcp_voxelize (its first pass over the rows is then already done). */
/* Round 5 (DiscoNet under hipGraph): which (agent, frame) maps of a BEV maker exist, decided on the device instead of by the host read of
 * bev_maker.py:156.  pcp_agent_frame_live: live[a * batch + b] (64 * batch int32, 16-byte aligned) = 1 iff agent a (id in `agent_col`,
 * 0..63) holds a row anywhere in the batch (an agent without rows is skipped by the reference's maker; which frames of an agent exist is
 * metadata, known on the host).  pcp_zero_maps_unless: `maps` =
 * n_maps (<= 256) consecutive maps of map_elems floats; map m is zero-filled unless live[flag_index_host[m]] != 0 (flag_index_host[m] < 0:
 * left alone).  Both capturable: no host read. */
int pcp_agent_frame_live(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, int32_t batch, int32_t *live,
                         void *stream);
int pcp_zero_maps_unless(float *maps, int64_t map_elems, int32_t n_maps, const int32_t *flag_index_host, const int32_t *live,
                         void *stream);
size_t pcp_select_transform_compact_workspace_bytes(int64_t n, int32_t n_slots);
int pcp_select_transform_compact(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, int32_t n_slots,
                                 const float *agents_host, int32_t batch, const float *poses_host, const uint8_t *present_host,
                                 float *out, int64_t out_rows, void *workspace, size_t workspace_bytes, int32_t *slot_start,
                                 const pcp_grid_t *vox_grid, void *vox_workspace, size_t vox_workspace_bytes, void *stream);
/* Round 6: the same with the pose table (n_slots * batch x 12 floats) and the presence flags (n_slots * batch bytes) in DEVICE memory: under a
 * captured hipGraph they are refreshed by the host before a replay (which agents exist at all -- agents_host, n_slots -- stays structure). */
int pcp_select_transform_compact_dev(const float *points, int64_t n, int32_t row_stride, int32_t agent_col, int32_t n_slots,
                                     const float *agents_host, int32_t batch, const float *poses_dev, const uint8_t *present_dev,
                                     float *out, int64_t out_rows, void *workspace, size_t workspace_bytes, int32_t *slot_start,
                                     const pcp_grid_t *vox_grid, void *vox_workspace, size_t vox_workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * SURVEY 8(f) rows 1-2  lately-fusion exchange: producer rows and ego-side MoDAR ingestion.
 * pcp_points_in_boxes   replaces pcdet/ops/roiaware_pool3d/src/roiaware_pool3d_api (points_in_boxes_gpu,
 *   roiaware_pool3d_kernel.cu:23-36,313-336): boxes (B, n_boxes, box_stride >= 7) [x,y,z,dx,dy,dz,heading,...], points
 *   (B, n_points, point_stride >= 3); box_idx (B, n_points) int32 = first containing box or -1.
 * pcp_hunter_foreground_rows   replaces pcdet/models/bev_layers/hunter_jr.py:377-397: rows i with sigmoid(head[i, 0]) < thresh_bg
 *   are written IN ORDER as [points[i, 1:], sigmoid(head[i, 0:3]), head[i, 3:6]] (row_stride - 1 + 6 columns) with their frame
 *   index in row_batch; count (1,) int32 on the device.  rows / row_batch have capacity n.
 * pcp_modar_ingest   replaces pcdet/datasets/v2x_sim/v2x_sim_dataset_ego.py:196-232: modar (n, 9) [box7, score, label];
 *   foreground (m, cols) [x,y,z, ..., flow(3) in the last three columns] or NULL; target_se3_lidar_host: 12 doubles (row-major
 *   3x4, HOST); rows (n, 13) = [x,y,z, 0, 0, dx,dy,dz, heading, score, label, max_sweep_idx, -1].
 * ------------------------------------------------------------------------------------------------------------------ */
int pcp_points_in_boxes(const float *boxes, int32_t batch, int32_t n_boxes, int32_t box_stride, const float *points, int32_t n_points,
                        int32_t point_stride, int32_t *box_idx, void *stream);
size_t pcp_hunter_foreground_workspace_bytes(int64_t n);
int pcp_hunter_foreground_rows(const float *points, int64_t n, int32_t row_stride, const float *head, int32_t ld_head, float thresh_bg,
                               void *workspace, size_t workspace_bytes, float *rows, int32_t *row_batch, int32_t *count, void *stream);
int pcp_modar_ingest(const float *modar, int32_t n_modar, const float *foreground, int32_t n_foreground, int32_t foreground_cols,
                     const double *target_se3_lidar_host, float max_sweep_idx, float *rows, void *stream);

/* The same ingestion for ALL remote agents of ALL frames at once, driven by device-resident counts (no host synchronisation; config 3 of
 * BASELINE.json on one GPU: pcdet/models/lately_chain.py).  Group g = one (frame, remote agent) pair.
 * det_*: (groups, det_max, 7) / (groups, det_max) float32 / (groups, det_max) int64 (1-based labels) / (groups,) int32 -- exactly what
 *        pcp_gather_detections leaves for the stacked remote pass;
 * foreground: (>= *foreground_count, cols) rows of pcp_hunter_foreground_rows over the stacked pass, foreground_group their frame-slot
 *        index (ascending: the rows keep the original point order), foreground_count a device int32, max_foreground the row capacity;
 * poses: device (groups, 12) float64 row-major 3x4 target_se3_lidar; max_sweep_idx: device (groups,) float32; frame_of_group: device
 *        (groups,) int32 = the ego frame whose cloud receives the rows;
 * rows:  (groups * det_max, 14) = [frame index | the 13 columns of pcp_modar_ingest]; padding slots carry frame index -1 (pcp_voxelize
 *        drops them).  Same arithmetic as pcp_modar_ingest (first containing box wins a point, float32 mean * 2, float64 pose). */
size_t pcp_modar_ingest_batched_workspace_bytes(int32_t groups, int64_t max_foreground);
int pcp_modar_ingest_batched(const float *det_boxes, const float *det_scores, const int64_t *det_labels, const int32_t *det_count,
                             int32_t groups, int32_t det_max, const float *foreground, int32_t foreground_cols,
                             const int32_t *foreground_group, const int32_t *foreground_count, int64_t max_foreground,
                             const double *poses, const float *max_sweep_idx, const int32_t *frame_of_group, void *workspace,
                             size_t workspace_bytes, float *rows, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * SURVEY 8(f) row 3  AnchorHeadSingle: box decoding + class-agnostic candidate selection.
 * Replaces pcdet/models/dense_heads/anchor_head_template.py:225-272 (generate_predicted_boxes), pcdet/utils/box_coder_utils.py:46-78
 *          (ResidualCoder.decode_torch), common_utils.limit_period, and detector3d_template.py:262-326 up to the NMS call
 *          (sigmoid, max over classes, score mask >=, torch.topk).
 * head: (B, H, W, ld) NHWC map of the fused 1x1 convs: class logits at channel ch_cls + a * num_class + c, box code at
 *   ch_box + a * 7 + j, direction logits at ch_dir + a * num_dir_bins + k (num_dir_bins = 0: no direction classifier).
 * anchors: (H * W * anchors_per_loc, 7) in the reference's order (y, x, size, rotation).
 * pcp_anchor_decode outputs, N = H * W * anchors_per_loc: boxes (B, N, 7) = batch_box_preds; cls_logits (B, N, num_class) =
 *   batch_cls_preds; score_keys (B, N) uint32 = bits(sigmoid(max logit)) + 1, or 0 when below score_thresh; labels (B, N) int32 arg-max.
 * pcp_topk_boxes: per frame the k (<= 4096) largest keys, descending, ties to the lower index -> gathered boxes (B, k, 7), scores,
 *   labels, source indices, count (B,) = min(k, number of non-zero keys).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t batch, h, w, ld;
  int32_t anchors_per_loc, num_class, num_dir_bins;
  int32_t ch_cls, ch_box, ch_dir;
  float dir_offset, dir_limit_offset, dir_period;      /* DIR_OFFSET, DIR_LIMIT_OFFSET, 2 pi / NUM_DIR_BINS */
  int32_t use_score_thresh;
  float score_thresh;
} pcp_anchor_t;

int pcp_anchor_decode(const pcp_anchor_t *desc, const float *head, const float *anchors, float *boxes, float *cls_logits,
                      uint32_t *score_keys, int32_t *labels, void *stream);
int pcp_topk_boxes(const uint32_t *score_keys, const int32_t *labels, const float *boxes, int32_t batch, int64_t n, int32_t k,
                   float *out_boxes, float *out_scores, int32_t *out_labels, int32_t *out_index, int32_t *count, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCP_HIP_H */
