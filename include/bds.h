/*
 * bds.h -- C ABI of libbds.so: the MI355X (gfx950) hot path of bilateral-driving.
 *
 * The reference (BigCiLeng/bilateral-driving) is pure Python; its hot path is reached through
 * two Python import surfaces, not an FFI.  The functions declared here are the device entry
 * points those surfaces bind to (through ctypes, see bilateral_driving_amd/_lib.py and
 * INTEGRATION.md).  Each declaration cites the reference interface it serves
 * (paths relative to /root/reference/project).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major memory (fp32 unless noted),
 *     owned by the caller (the torch caching allocator); nothing is allocated inside;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), except
 *     bds_isect_prepare which must hand the intersection count back to the host;
 *   - return value: BDS_OK or a negative BDS_E* code; never throws, never exits;
 *   - C = cameras, N = Gaussians, M = tile intersections, H/W = image size, tile = 16.
 */
#ifndef BDS_H
#define BDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDS_ABI_VERSION 2   /* 2: round 5 changed exported signatures (split_len / split_cap, error_pinned) + the round-6 entries */

#define BDS_OK 0
#define BDS_EINVAL (-1)      /* null / misaligned pointer, bad shape or unsupported parameter */
#define BDS_EWORKSPACE (-2)  /* workspace too small */
#define BDS_ELAUNCH (-3)     /* hipGetLastError() != hipSuccess after a launch / memcpy */
#define BDS_ECAPACITY (-4)   /* bds_isect_tiles: more intersections than the caller's buffers hold (see there) */

typedef void *bds_stream_t;

int bds_abi_version(void);
const char *bds_strerror(int code);

/* Test hooks that force the large-input fallback paths of the tile stage on small inputs (there is ONE kernel per
 * operation; these select which size regime a call is treated as).  which:
 * 0 = device-count tile stage: 1 [default] = launches behind the compaction sized by the visible-entry CAPACITY, 0 = by C*N;
 * 4 = depth ordering of the visible entries: 1 [default] = the two-launch radix passes used up to 8.4 M (camera, Gaussian)
 *     entries; 0 = the generic histogram / scan / scatter passes that larger inputs take;
 * 6 = tile lists: 1 [default] = packed 32-bit entries (tile << rank_bits | depth rank) whenever the visible count fits the
 *     rank bits; 0 = the (tile key, id) pair lists that larger visible counts take.
 * 3 = profiling only: ablation mask of the bilateral backward;
 * 7 = bilateral transform, bit mask [default 3]: bit 0 = the cell-aligned kernels (csrc/bilagrid_cells.hip) wherever a level
 *     qualifies (one grid per level); bit 1 = the pyramid forward as one pass over the image (csrc/bilagrid_tile.hip) when every
 *     factor is a power of two >= 2 dividing the image; bit 2 = do NOT defer the backward's last stage to the compositor
 *     (bds_bilagrid_ms_ed_bwd_deferrable returns 0); 0 = the general kernels everywhere (what levels averaged over several
 *     grids always take).  Same results.
 * Other indices are unused. */
int bds_set_option(int which, int value);
int bds_get_option(int which);

/* Timing marks (measurement plumbing, no reference counterpart): HIP events that are recorded as event-record NODES when the
 * stream is being captured into a hipGraph, so that every replay re-records them and a kernel inside a captured view can be
 * bracketed (bench.py's roofline kernel).  bds_timer_elapsed_ms waits for `stop`; negative = not available. */
void *bds_timer_create(void);
int bds_timer_destroy(void *timer);
int bds_timer_mark(void *timer, bds_stream_t stream);
float bds_timer_elapsed_ms(void *start, void *stop);
int bds_last_hip_error(void);   /* HIP error code of the last failed runtime call of the timing-mark entry points (diagnostics) */

/* ---- spherical harmonics ----------------------------------------------------------------
 * gsplat.cuda._wrapper.spherical_harmonics(degrees_to_use, dirs, coeffs, masks=None)
 * imported at models/gaussians/basics.py:15, called at models/gaussians/vanilla.py:388
 * (and pvg.py:403, deformgs.py:142, nodes/rigid.py:462, deformable.py:83, smpl.py:364).
 * dirs [n,3] (normalised inside), coeffs [n,K,3], masks [n] uint8 or NULL, out [n,3].
 * Only the first (deg+1)^2 of the K bases are read.  v_dirs may be NULL. */
int bds_sh_fwd(int64_t n, int K, int deg, const float *dirs, const float *coeffs, const uint8_t *masks, float *out,
               bds_stream_t stream);
int bds_sh_bwd(int64_t n, int K, int deg, const float *dirs, const float *coeffs, const uint8_t *masks,
               const float *v_out, float *v_coeffs, float *v_dirs, bds_stream_t stream);

/* ---- projection -------------------------------------------------------------------------
 * first stage of gsplat.rendering.rasterization (called at models/trainers/base.py:393-408,
 * 811-826): quat+scale -> 3D covariance, world->camera, perspective Jacobian, eps2d blur,
 * conic, 3-sigma radius, near/far/screen culling.
 * means [N,3] quats [N,4] (wxyz, normalised inside) scales [N,3] viewmats [C,4,4] Ks [C,3,3]
 * -> radii [C,N] i32 (0 = culled), means2d [C,N,2], depths [C,N], conics [C,N,3],
 *    compensations [C,N] or NULL ("antialiased" mode only).  Culled entries are zero-filled. */
int bds_project_fwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                    const float *viewmats, const float *Ks, int W, int H, float eps2d, float near_plane,
                    float far_plane, float radius_clip, int32_t *radii, float *means2d, float *depths, float *conics,
                    float *compensations, bds_stream_t stream);
/* v_means [N,3] v_quats [N,4] v_scales [N,3] are written (summed over cameras);
 * v_viewmats [C,4,4] (NULL = not needed) is zeroed and accumulated inside
 * (learnable camera poses: models/trainers/base.py:328-329,399).
 * v_compensations / compensations may be NULL. */
int bds_project_bwd(int C, int64_t N, const float *means, const float *quats, const float *scales,
                    const float *viewmats, const float *Ks, int W, int H, float eps2d, const int32_t *radii,
                    const float *conics, const float *compensations, const float *v_means2d, const float *v_depths,
                    const float *v_conics, const float *v_compensations, float *v_means, float *v_quats,
                    float *v_scales, float *v_viewmats, bds_stream_t stream);

/* ---- tile intersection + (tile|depth) ordering -------------------------------------------
 * isect_tiles + radix sort + isect_offset_encode stages of gsplat.rendering.rasterization.
 * Two calls because the host must size the [M] outputs:
 *   bds_isect_prepare : depth-orders the visible Gaussians, counts their tiles in that order (per-member
 *                       records and per-256-member totals stay in `ws`); synchronises `stream` and
 *                       returns M in *n_isects.
 *   bds_isect_build   : emits (camera*tiles+tile, id) pairs in depth order, stable-sorts them
 *                       by tile, writes flatten_ids [M] i32 (= cam*N+gaussian), isect_offsets
 *                       [C,th,tw] i32 and, if not NULL, isect_ids [M] i64
 *                       (cam|tile id << 32 | fp32 depth bits), identical to sorting the 64-bit
 *                       keys directly.
 * `ws` (bds_isect_prepare_workspace_bytes) must stay alive and untouched between the two calls;
 * `ws2` (bds_isect_build_workspace_bytes) is scratch for build.  M must be < 2^31. */
size_t bds_isect_prepare_workspace_bytes(int C, int64_t N);
/* Byte offset, inside the prepare workspace, of the ascending id list of the visible entries (int32 [n_visible], compact mode):
 * a caller that keeps `ws` alive reads the list in place and passes visible_ids = NULL to bds_isect_build (no copy). */
size_t bds_isect_visible_ids_offset(int C, int64_t N);
size_t bds_isect_build_workspace_bytes(int C, int64_t N, int64_t M);
/* conics [C,N,3] + opacities [C,N] (both or neither): when given, (tile, Gaussian) pairs in which no
 * pixel centre can reach alpha >= 1/255 are dropped ("exact tile culling"): rendered images and
 * gradients are unchanged, M shrinks; when NULL the lists are gsplat's bounding-square lists.
 * tiles_per_gauss [C,N] i32 may be NULL (the per-Gaussian counts are then not written). */
/* n_visible (may be NULL): number of (camera, Gaussian) entries with radii > 0.  Handing it to bds_isect_build
 * (or -1 when unknown) lets the build use PACKED lists when every entry's depth rank fits next to the tile key in one
 * 32-bit word (rank < 2^(32 - bits(C*tiles))): the tile sort then moves 4 bytes per entry instead of 8.  Same outputs. */
int bds_isect_prepare(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                      const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                      int32_t *tiles_per_gauss, void *ws, size_t ws_bytes, int64_t *n_isects, int64_t *n_visible, int compact,
                      bds_stream_t stream);
int bds_isect_build(int C, int64_t N, int64_t M, int64_t n_visible, const float *means2d, const int32_t *radii,
                    const float *depths,
                    const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h, const void *ws,
                    size_t ws_bytes, void *ws2,
                    size_t ws2_bytes, int64_t *isect_ids, int32_t *flatten_ids, int32_t *isect_offsets,
                    int32_t *visible_ids, int compact, bds_stream_t stream);
/* COMPACT lists (no reference counterpart).  compact != 0 (the same value in prepare and build; isect_ids must be NULL): every
 * intersection's list value is the entry's POSITION in the ascending list of the visible entries instead of its id cam*N+g
 * (same list order).  visible_ids (may be NULL; needs n_visible >= 0) receives that ascending list (position -> id) in either
 * mode.  Compact lists address splat records packed through visible_ids (bds_splat_pack), the compositor's gradient records
 * come back in that order, and everything downstream walks the visible ~15 % of the scene in memory order.
 * Asynchronous prepare: same work, but instead of synchronising it copies {M, n_visible} into `counts_pinned`
 * (int64[2], page-locked host memory) and records `event` (a hipEvent_t) on the stream.  The caller may enqueue
 * independent work, then waits for the event, reads the counts and calls bds_isect_build: the GPU keeps running that
 * work while the host sizes the lists. */
int bds_isect_prepare_async(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                            const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                            int32_t *tiles_per_gauss, void *ws, size_t ws_bytes, int64_t *counts_pinned, void *event,
                            int compact, bds_stream_t stream);
/* One-call form: bds_isect_prepare and then, without returning to the caller in between, bds_isect_build into
 * buffers sized for an EXPECTED count (flatten_ids / isect_ids hold flatten_capacity entries, ws2 is
 * bds_isect_build_workspace_bytes(C, N, flatten_capacity)).  M <= flatten_capacity: BDS_OK, *n_isects = M, lists
 * written (entries [M, capacity) untouched).  Otherwise BDS_ECAPACITY, *n_isects = M, nothing built: allocate for M
 * and call bds_isect_build (ws is intact).  Saves the host's allocation work between the two stages, during which
 * the GPU idles. */
int bds_isect_tiles(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                    const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                    int32_t *tiles_per_gauss, void *ws, size_t ws_bytes, void *ws2, size_t ws2_bytes,
                    int64_t flatten_capacity, int64_t *isect_ids, int32_t *flatten_ids, int32_t *isect_offsets,
                    int64_t *n_isects, int64_t *n_visible, bds_stream_t stream);

/* ---- alpha compositing -------------------------------------------------------------------
 * rasterize_to_pixels stage of gsplat.rendering.rasterization; outputs consumed at
 * models/trainers/base.py:409-419 (render, alphas) and :280-297 (means2d.absgrad).
 *
 * The compositor reads SPLAT RECORDS: 12 floats (48 bytes, 16-byte aligned) per list-addressable entry,
 *     mean2d.x, mean2d.y, ea, eb | ec, opacity, colour0, colour1 | colour2, colour3, 0, radius (int32 bits)
 * with (ea, eb, ec) = -log2(e) * (a/2, b, c/2) of the conic (a, b, c): alpha = opacity * 2^(ea dx^2 + eb dx dy + ec dy^2).
 * bds_splat_pack builds them from the per-entry arrays means2d [n,2] conics [n,3] colors [n,CH] opacities [n]: record r is
 * entry ids[r], or entry r when ids == NULL.  The per-tile lists (`flatten`) hold RECORD indices: cam*N + g for records in
 * array order (gsplat's flatten_ids), or depth ranks of the visible entries for records packed through a sorted id list.
 * CH in {1,3,4}; backgrounds [C,CH] or NULL -> render [C,H,W,CH], alphas [C,H,W], last_ids [C,H,W] i32.
 *
 * COARSE LISTS (no reference counterpart): list_tile_size is the tile size the lists were BUILT for (bds_isect_* called with it:
 * isect_offsets [C, ceil(H/list_tile_size), ceil(W/list_tile_size)]) -- tile_size (16, gsplat's lists) or a multiple of it.  With a
 * multiple, the tile stage emits and sorts one pair per (list tile, Gaussian) instead of one per (16-px tile, Gaussian), and every
 * 16 x 16 compositing wave filters the candidates of its list tile as it stages them, by the two conditions that define a pair of
 * gsplat's 16-px lists: the tile lies in the Gaussian's bounding square (radius slot of the record) and one of its pixel centres
 * reaches alpha >= 1/255 (the tile stage's exact span test).  Records packed WITH radii therefore give exactly the render / alphas /
 * gradients of 16-px lists (the fused view's use); records packed without (radii == NULL: unbounded square) give gsplat's semantics
 * for tile_size = list_tile_size, where a splat is clipped to its bounding square at that tile granularity (the rasterization()
 * API's tile_size argument).  last_ids holds positions in the coarse list. */
#define BDS_SPLAT_RECORD_FLOATS 12
#define BDS_GRAD_RECORD_FLOATS 16
/* Glue of gsplat's rasterization() as ONE node for one camera (rendering._RasterizeView; models/trainers/base.py:393-408):
 * records of the visible entries straight from post-activation colours [N,3] + depths [N] (channel 3 = depth, "RGB+ED");
 * the expected-depth normalise out4 = (rgb, D / max(alpha, 1e-10)) and its backward, which also widens a 3-channel ("RGB") image
 * gradient to the compositor's four channels: v_render4 = (v_out.rgb, v_out.d / max(alpha, 1e-10) | 0), v_alphas = v_alphas_in
 * (may be NULL) - [alpha >= 1e-10] D v_out.d / max(alpha, 1e-10)^2.  v_out: [P, channels], may be NULL (zeros). */
int bds_splat_pack_rgbd(int64_t n, const int32_t *ids, const float *means2d, const float *conics, const float *colors3,
                        const float *depths, const float *opacities, const int32_t *radii, float *records, bds_stream_t stream);
int bds_expected_depth_fwd(int64_t P, const float *render4, const float *alphas, float *out4, bds_stream_t stream);
int bds_expected_depth_bwd(int64_t P, int channels, int expected_depth, const float *render4, const float *alphas, const float *v_out,
                           const float *v_alphas_in, float *v_render4, float *v_alphas, bds_stream_t stream);
/* The same with the image as TWO arrays, rgb3 [P,3] and depth1 [P,1] (expected_depth = 0: the composited depth as it is): the
 * reference's trainer splits the render at once (models/trainers/base.py:409-419: torch.split(renders, [3, 1], dim=-1)); handed out as
 * two outputs of the rasterization node (rendering.SplitRender) that split costs no copy forward and no slice backward.
 * v_rgb3 / v_depth1 / v_alphas_in may be NULL (zero). */
int bds_expected_depth_split_fwd(int64_t P, int expected_depth, const float *render4, const float *alphas, float *rgb3, float *depth1,
                                 bds_stream_t stream);
int bds_expected_depth_split_bwd(int64_t P, int expected_depth, const float *render4, const float *alphas, const float *v_rgb3,
                                 const float *v_depth1, const float *v_alphas_in, float *v_render4, float *v_alphas, bds_stream_t stream);
int bds_splat_pack(int64_t n, int CH, const int32_t *ids, const float *means2d, const float *conics, const float *colors,
                   const float *opacities, const int32_t *radii /* [n entries] or NULL */, float *records, bds_stream_t stream);
/* Splat records of the fused view with the SH colour evaluated on the way (vanilla.py:383-389: SH of normalise(means - cam_pos), + 0.5,
 * clamp to [0,1]; channel 3 = depth): record r = visible Gaussian ids[r]; coeffs [N,K,3] with K*3 a multiple of 4 and 16-byte aligned
 * rows; sh_rgb [n,3] receives the un-clamped colours in list order (for bds_sh_view_bwd_list with sh_rgb_by_rank).  Replaces
 * bds_sh_view_fwd + bds_splat_pack on that path: the colours of the ~85 % culled Gaussians are never evaluated or stored. */
int bds_splat_pack_sh(int64_t n, const int32_t *ids, int K, int degrees_to_use, const float *means, const float *cam_pos,
                      const float *coeffs, const float *means2d, const float *conics, const float *depths, const float *opacities,
                      const int32_t *radii, float *records, float *sh_rgb, bds_stream_t stream);
/* The same with the coefficients where the reference's Gaussian classes hold them (models/gaussians/vanilla.py:96-104: `_features_dc`
 * [N,3] and `_features_rest` [N,K-1,3], two parameters with their own learning rates; :382 concatenates them on every call --
 * 192 bytes per Gaussian written and read back before anything is culled): band 0 from coeffs_dc, bands 1.. from coeffs_rest
 * (ignored for K = 1), rows of 4-byte alignment, visible rows only. */
int bds_splat_pack_sh_split(int64_t n, const int32_t *ids, int K, int degrees_to_use, const float *means, const float *cam_pos,
                            const float *coeffs_dc, const float *coeffs_rest, const float *means2d, const float *conics,
                            const float *depths, const float *opacities, const int32_t *radii, float *records, float *sh_rgb,
                            bds_stream_t stream);
int bds_rasterize_fwd(int C, int64_t n_records, int64_t M, int CH, const float *records, const float *backgrounds, int W, int H,
                      int tile_size, int list_tile_size, int tile_w, int tile_h, const int32_t *isect_offsets,
                      const int32_t *flatten, float *render, float *alphas, float *t_final, int32_t *last_ids, bds_stream_t stream);
/* (t_final [C,H,W], optional -- NULL: not written: every pixel's final transmittance ITSELF.  alphas = 1 - T rounds the low bits of a
 * small T away (T = 1e-4: 6e-4 relative), and the backward divides its way up from T: handed the same buffer, the backward starts
 * from the exact value -- element-wise gradient errors of dense scenes drop from ~1e-2 to the fp32 formulation's ~1e-3,
 * profiles/r09_gs_gradient_errors.json.  gsplat's backward starts from 1 - render_alphas.) */
/* Backward into GRADIENT RECORDS v_records [n_records, 16] (64-byte stride, zero-filled by the caller, accumulated with
 * atomics), in the units of the un-scaled inputs:
 *     0-3 d/d colour | 4-6 d/d conic (a, b, c) | 7-8 d/d mean2d | 9-10 sum over pixels of |d/d mean2d| (absgrad != 0) |
 *     11 d/d opacity | 12-15 unused.
 * tile_order may be NULL (tiles are taken in image order, one contiguous band per XCD) or the schedule written by
 * bds_rasterize_bwd_schedule. */
int bds_rasterize_bwd(int C, int64_t n_records, int64_t M, int CH, const float *records, const float *backgrounds, int W, int H,
                      int tile_size, int list_tile_size, int tile_w, int tile_h, const int32_t *isect_offsets,
                      const int32_t *flatten, const float *alphas, const float *t_final, const int32_t *last_ids, const float *v_render,
                      const float *v_alphas, float *v_records, int absgrad, const int32_t *tile_order, bds_stream_t stream);
/* Launch schedule for bds_rasterize_bwd (no reference counterpart; results do not depend on it).  One wave owns a
 * tile and the chip holds only about two rounds of tiles, so the launch ends with a tail of long tiles that started
 * late.  After the forward pass each tile's visited length is known exactly (max last_id - list start); this call
 * orders every XCD's contiguous range of tiles longest-first.  tile_order: int32[bds_rasterize_schedule_ints(C, tile_w, tile_h)]:
 * word 0 tags the form (1 = sorted: the order follows, then scratch; 0 = binned, written by bds_rasterize_fwd_dev), which the
 * backward kernels read -- hand the buffer over as it is. */
int64_t bds_rasterize_schedule_ints(int C, int tile_w, int tile_h);
int bds_rasterize_bwd_schedule(int C, int W, int H, int tile_size, int list_tile_size, int tile_w, int tile_h, const int32_t *isect_offsets,
                               const int32_t *last_ids, int32_t *tile_order, bds_stream_t stream);

/* ---- bilateral grid ----------------------------------------------------------------------
 * Point slice: bilateral/lib_bilagrid.py:317-368 BilateralGrid.forward (F.grid_sample,
 * trilinear, align_corners=True, padding_mode="border") used by slice() :171-230.
 * grid [12,L,gy,gx]; xy [P,2] in [0,1]; rgb [P,3] (guidance = BT.601 gray) -> affine [P,12].
 * bwd accumulates into v_grid (caller zero-fills) and writes v_rgb [P,3] (guidance route only). */
int bds_bilagrid_slice_fwd(int64_t P, const float *grid, int gx, int gy, int gl, const float *xy, const float *rgb,
                           float *affine, bds_stream_t stream);
int bds_bilagrid_slice_bwd(int64_t P, const float *grid, int gx, int gy, int gl, const float *xy, const float *rgb,
                           const float *v_affine, float *v_grid, float *v_rgb, bds_stream_t stream);

/* The same slice for FEATURE grids of any channel count NC (bilateral/lib_bilagrid.py:370-461 NeuralBilateralGrid.forward,
 * used by slice_feature :232-253): grid [NC,L,gy,gx] -> out [P,NC]; v_grid accumulated, v_rgb written (guidance route). */
int bds_bilagrid_slice_feat_fwd(int64_t P, int NC, const float *grid, int gx, int gy, int gl, const float *xy,
                                const float *rgb, float *out, bds_stream_t stream);
int bds_bilagrid_slice_feat_bwd(int64_t P, int NC, const float *grid, int gx, int gy, int gl, const float *xy,
                                const float *rgb, const float *v_out, float *v_grid, float *v_rgb, bds_stream_t stream);

/* The same slice over a whole IMAGE: xy is implied -- pixel (x, y) samples at (linspace(0,1,W)[x], linspace(0,1,H)[y]), which is what
 * the neural modules slice at (models/modules.py:643-650, 728-760: torch.meshgrid of two linspaces) -- so a workgroup can stage the
 * two grid rows a pixel row touches in LDS (the band [NC][gl][2][gx]) and scatter the gradient there.  rgb [H,W,3] -> out [H,W,NC]
 * (16-byte aligned when NC % 4 == 0).  bwd ACCUMULATES into v_grid [NC,gl,gy,gx] (caller zeroes; may be NULL) and writes v_rgb [H,W,3]
 * (the guidance route; may be NULL).  bds_bilagrid_slice_feat_image_ok(NC, gx, gy, gl) != 0 when the band fits (NC*gl*2*gx <= 6144
 * floats: the shipped 16x16x8 grid with 24 features exactly); otherwise the calls return BDS_EINVAL and the point form applies. */
int bds_bilagrid_slice_feat_image_ok(int NC, int gx, int gy, int gl);
int bds_bilagrid_slice_feat_image_fwd(int H, int W, int NC, const float *grid, int gx, int gy, int gl, const float *rgb, float *out,
                                      bds_stream_t stream);
int bds_bilagrid_slice_feat_image_bwd(int H, int W, int NC, const float *grid, int gx, int gy, int gl, const float *rgb,
                                      const float *v_out, float *v_grid, float *v_rgb, bds_stream_t stream);

/* Fused image transform: models/modules.py:505-522 MultiScaleBilateralAffineTransform.forward
 * (train branch: get_sample_grid :494-504, slice, fill_matrix_res :409-420), the single-scale
 * BilateralAffineTransform.forward :317-335 (factor 1), the sequential application at
 * models/trainers/scene_graph.py:95-98,112-117, and optionally the clamp + sky blend in front of
 * it (trainers/base.py:417, scene_graph.py:292-294).
 *
 * Level l: grid [n_avg, 12, L, gy, gx] (n_avg = 1 for training; > 1 averages the low-res
 * slices of neighbouring frames' grids = the test branch, modules.py:523-535).
 * If sky != NULL the input colour is clamp(rgb, max=1) + sky * (1 - alpha). */
#define BDS_MAX_LEVELS 8
typedef struct {
  const float *grid; /* [n_avg,12,gl,gy,gx] */
  float *v_grid;     /* same shape, accumulated (caller zero-fills); may be NULL in fwd */
  int32_t gx, gy, gl, factor, n_avg;
} bds_bilagrid_level_t;

size_t bds_bilagrid_ms_workspace_bytes(int nlevels, const bds_bilagrid_level_t *levels, int H, int W);
/* ws keeps the low-resolution affine maps (fwd -> bwd). affine_out (NULL or nlevels x [H,W,12]
 * pointers) receives the full-resolution per-level maps the reference module returns. */
int bds_bilagrid_ms_fwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *rgb,
                        const float *alpha, const float *sky, void *ws, size_t ws_bytes, float *rgb_out,
                        float *const *affine_out, bds_stream_t stream);
/* v_rgb [H,W,3] written; v_alpha [H,W], v_sky [H,W,3] written when sky != NULL. */
int bds_bilagrid_ms_bwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *rgb,
                        const float *alpha, const float *sky, void *ws, size_t ws_bytes, const float *v_rgb_out,
                        float *v_rgb, float *v_alpha, float *v_sky, bds_stream_t stream);

/* TV regulariser: bilateral/lib_bilagrid.py:152-168 total_variation_loss on [n,12,L,gy,gx].
 * tv_out [1] is accumulated (caller zero-fills) with weight*tv; v_grids += weight*v_tv*dtv/dgrid. */
int bds_bilagrid_tv_fwd(int64_t n, int gx, int gy, int gl, const float *grids, float weight, float *tv_out,
                        bds_stream_t stream);
int bds_bilagrid_tv_bwd(int64_t n, int gx, int gy, int gl, const float *grids, float weight, const float *v_tv,
                        float *v_grids, bds_stream_t stream);
/* ... and for grids with `channels` != 12 (NeuralBilateralGrid.tv_loss): grids [n, channels, gl, gy, gx] */
int bds_grid_tv_fwd(int64_t n, int channels, int gx, int gy, int gl, const float *grids, float weight, float *tv_out,
                    bds_stream_t stream);
int bds_grid_tv_bwd(int64_t n, int channels, int gx, int gy, int gl, const float *grids, float weight, const float *v_tv,
                    float *v_grids, bds_stream_t stream);
/* The same for all levels of a multi-scale transform in ONE launch each way: levels[l].grid / v_grid [n_avg,12,gl,gy,gx]
 * (n_avg = number of images, factor unused), weights[l] as above; tv_out accumulates sum_l weights[l] * tv_l. */
int bds_bilagrid_tv_ms_fwd(int nlevels, const bds_bilagrid_level_t *levels, const float *weights, float *tv_out,
                           bds_stream_t stream);
int bds_bilagrid_tv_ms_bwd(int nlevels, const bds_bilagrid_level_t *levels, const float *weights, const float *v_tv,
                           bds_stream_t stream);

/* L1 + TV training loss of the direct step, value and gradient in ONE launch: the loss = mean|a - b| + sum_l weights[l] *
 * TV(levels[l].grid) is ADDED, workgroup by workgroup, to loss_out [loss_slots * BDS_LOSS_SLOT_STRIDE] (zeroed by the caller;
 * loss_slots a power of two; the value is the sum of the slots' first floats -- thousands of float atomics on ONE address serialise
 * at ~8 ns each); v_a [n] = sign(a - b) * v_loss / n; levels[l].v_grid (may be NULL) += v_loss * d(weights[l] * TV)/d(grid), with
 * atomics (concurrent views may add to the same slices).  nlevels may be 0.
 * (models/trainers/base.py:518-565 rgb term + the `losses.affine` TV term, models/modules.py:445,466-472.) */
#define BDS_LOSS_SLOT_STRIDE 64
int bds_l1_tv_train(int64_t n, const float *a, const float *b, int nlevels, const bds_bilagrid_level_t *levels, const float *weights,
                    float v_loss, float *loss_out, int loss_slots, float *v_a, bds_stream_t stream);

/* ---- one-view forms for the fused training step ------------------------------------------------------
 * The per-Gaussian glue of one reference iteration folded into the two streaming kernels that sit next to it
 * (C = 1; same arithmetic as the general forms above):
 *   project_view: scales = exp(log_scales), opacities = sigmoid(logits) (models/gaussians/vanilla.py:393-394) are
 *     computed by the projection (and returned: the compositor and the backward read them); the backward returns the
 *     gradients of the raw parameters and reads only the radius of a culled Gaussian.
 *   sh_view: view direction = means - cam_pos (vanilla.py:384, detached), visibility = radii > 0, output packed for
 *     the RGB+ED compositor as colors [N,4] = (clamp(sh + 0.5, 0, 1), depth) (vanilla.py:389); sh_rgb [N,3] keeps the
 *     un-clamped value for the backward. */
/* ROW FORM of the projection's outputs (optional, recognised by the addresses -- no argument says so): means2d, depths and conics may
 * be the COLUMNS of one 16-byte aligned [N,8] block of 32-byte rows {m2d.x, m2d.y, depth, radius (int bits) | conic a, b, c, opacity}:
 * pass means2d = block, depths = block + 2, conics = block + 4 (both or neither; separate arrays of more than one row can not have
 * these addresses).  bds_project_view_fwd / _prepare_fwd then write whole rows (radii [N] and opacities [N] are written as dense
 * arrays as well), and bds_isect_prepare* / bds_isect_build* / bds_splat_pack_sh* given the same three pointers (and opacities =
 * block + 7, or any dense [N] array of other opacities) gather ONE line per visible Gaussian instead of one per array. */
int bds_project_view_fwd(int64_t N, const float *means, const float *quats, const float *log_scales, const float *logits,
                         const float *viewmat, const float *K, int W, int H, float eps2d, float near_plane,
                         float far_plane, float radius_clip, float *scales, float *opacities, int32_t *radii,
                         float *means2d, float *depths, float *conics, bds_stream_t stream);
/* Block bounds: rows kept in spatial order (Morton order of the centres: the host side's densify.spatial_order) make every 256-row
 * block a small box, and a camera then rejects most of the ~85 % of the Gaussians it does not see a BLOCK at a time.
 * bds_gaussian_block_bounds: block_bounds [cdiv(N, 256), 8] = {lo.xyz, largest activated scale | hi.xyz, -} of the centres of rows
 * [256 b, 256 (b + 1)); recompute whenever means / log_scales changed (once per frame).  The _blocks forms of the two projections
 * below skip a block when no centre inside its box can come out visible (conservative: csrc/gs_math.h box_may_be_visible follows
 * the tests of the projection itself) -- its rows get a culled Gaussian's outputs (radius 0, zeros) without being read; scales /
 * opacities of such rows are not written.  Results are identical to the plain forms. */
int bds_gaussian_block_bounds(int64_t N, const float *means, const float *log_scales, float *block_bounds, bds_stream_t stream);
int bds_project_view_fwd_blocks(int64_t N, const float *means, const float *quats, const float *log_scales, const float *logits,
                                const float *viewmat, const float *K, int W, int H, float eps2d, float near_plane, float far_plane,
                                float radius_clip, float *scales, float *opacities, int32_t *radii, float *means2d, float *depths,
                                float *conics, const float *block_bounds, bds_stream_t stream);
int bds_project_view_prepare_fwd_blocks(int64_t N, const float *means, const float *quats, const float *log_scales, const float *logits,
                                        const float *viewmat, const float *K, int W, int H, float eps2d, float near_plane,
                                        float far_plane, float radius_clip, float *scales, float *opacities, int32_t *radii,
                                        float *means2d, float *depths, float *conics, int32_t *tiles_per_gauss, void *prep_ws,
                                        size_t prep_ws_bytes, const float *block_bounds, bds_stream_t stream);
/* bds_project_view_fwd that also does the first launch of the tile stage (device-count form, C = 1): the number of visible Gaussians
 * per 256-Gaussian workgroup is left in prep_ws (bds_isect_prepare_workspace_bytes(1, N)), the stage's sort tables and
 * tiles_per_gauss [N] (may be NULL) are cleared.  Follow with bds_isect_prepare_dev(..., compact | 2, ...) on the SAME workspace.
 * BDS_ECAPACITY when N is beyond the short sort path (use bds_project_view_fwd then). */
int bds_project_view_prepare_fwd(int64_t N, const float *means, const float *quats, const float *log_scales, const float *logits,
                                 const float *viewmat, const float *K, int W, int H, float eps2d, float near_plane, float far_plane,
                                 float radius_clip, float *scales, float *opacities, int32_t *radii, float *means2d, float *depths,
                                 float *conics, int32_t *tiles_per_gauss, void *prep_ws, size_t prep_ws_bytes, bds_stream_t stream);
int bds_sh_view_fwd(int64_t N, int K, int degrees_to_use, const float *means, const float *cam_pos, const float *coeffs,
                    const int32_t *radii, const float *depths, float *sh_rgb, float *colors, bds_stream_t stream);
/* Backward of the one-view forms over the VISIBLE entries only, list-driven (no reference counterpart; the reference's dense
 * backward writes zeros for every culled Gaussian).  ids [n_list] = depth-ordered ids of the visible entries (bds_isect_build:
 * visible_ids); v_records [n_list,16] = the compositor's gradient records of the same ranks (bds_rasterize_bwd with rank lists).
 * accumulate = 0 STORES the rows ids[.] of the gradient arrays (all other rows are the caller's business: zero-filled, or kept zero
 * with bds_view_grads_clear_list), accumulate = 1 ADDS to them (several views summed into one buffer before one exchange).
 *   sh_view_bwd_list     : v_coeffs [N,K,3] rows (colour clamp of vanilla.py:389 applied through sh_rgb).
 *   project_view_bwd_list: v_means [N,3] v_quats [N,4] v_log_scales [N,3] v_logits [N] rows; optionally (may be NULL)
 *       grad2d / absgrad2d [N,2]: the screen-space gradient and the sum over pixels of its absolute value scattered to the
 *       dense arrays models/trainers/base.py:280-297 reads (rows of culled Gaussians untouched);
 *       v_viewmat_slots [BDS_POSE_GRAD_SLOTS,4,4]: camera-pose gradient partials, ADDED to (the caller zero-fills them, e.g. as the
 *       tail of the gradient-record allocation; replicated accumulators: one address would serialise in L2); the sum over the slots
 *       is d(loss)/d(viewmat), models/trainers/base.py:328-329,399.
 * row_map (may be NULL) [N] i32: the parameter-gradient row of Gaussian g is row_map[g] instead of g -- the rows then land in a
 * compact exchange buffer (multi-GPU: the slot of g in the union of the ranks' visible sets) instead of the dense arrays. */
#define BDS_POSE_GRAD_SLOTS 64
/* ROW FORM of the four small parameter gradients (optional, recognised by the addresses like the projection's row form): v_means,
 * v_quats, v_log_scales, v_logits may be the columns of one 16-byte aligned [N,16] block of 64-byte rows {v_mean 3, v_logit | v_quat 4 |
 * v_log_scale 3, - | - - - -}: pass v_means = block, v_logits = block + 3, v_quats = block + 4, v_log_scales = block + 8.
 * bds_project_view_bwd_list* then update, and bds_view_grads_clear_list* clear, ONE line per visible Gaussian instead of four partly
 * used ones; bds_adam_step_rows reads such columns. */
/* sh_rgb: the un-clamped colours the forward left -- [N,3] indexed by Gaussian (bds_sh_view_fwd), or, with sh_rgb_by_rank != 0,
 * [n_list,3] in list order (bds_splat_pack_sh). */
int bds_sh_view_bwd_list(int64_t n_list, const int32_t *ids, int K, int degrees_to_use, const float *means, const float *cam_pos,
                         const float *sh_rgb, int sh_rgb_by_rank, const float *v_records, float *v_coeffs, const int32_t *row_map,
                         int accumulate, bds_stream_t stream);
/* ... into the split storage of bds_splat_pack_sh_split: v_coeffs_dc [N,3], v_coeffs_rest [N,K-1,3] (visible rows stored or added to). */
int bds_sh_view_bwd_list_split(int64_t n_list, const int32_t *ids, int K, int degrees_to_use, const float *means, const float *cam_pos,
                               const float *sh_rgb, int sh_rgb_by_rank, const float *v_records, float *v_coeffs_dc,
                               float *v_coeffs_rest, int accumulate, bds_stream_t stream);
/* NaN / Inf check of the tensors a Gaussian class hands to the rasterizer (models/gaussians/vanilla.py:407-412 raises ValueError per
 * tensor; two reductions and two host waits each there): ONE streaming launch over up to 8 tensors; bit t of *flags_dev (cleared
 * first) is set when tensors[t] (counts[t] floats) holds a non-finite value; flags_pinned (optional, page-locked) receives a copy
 * behind the launch, for the host to read after its next wait on the stream. */
int bds_nonfinite_flags(int n_tensors, const float *const *tensors, const int64_t *counts, uint32_t *flags_dev,
                        uint32_t *flags_pinned, bds_stream_t stream);
/* The same with a KIND per tensor (kinds [n_tensors], NULL = all 0): the bit says "the tensor's ACTIVATED value would hold a NaN / Inf"
 * for raw parameters whose activation runs inside a kernel (vanilla.py:393-395, checked at :407-412 on the activated tensors):
 * 0 plain (NaN, +-Inf) | 1 argument of exp (NaN, +Inf, x >= 88.72284: exp overflows; -Inf is fine) | 2 quaternion rows [n/4, 4],
 * 16-byte aligned (NaN / Inf components, or a row whose squared norm is 0 in fp32: 0/0, x/0) | 3 argument of sigmoid (NaN only). */
int bds_nonfinite_flags_kinds(int n_tensors, const float *const *tensors, const int64_t *counts, const int *kinds, uint32_t *flags_dev,
                              uint32_t *flags_pinned, bds_stream_t stream);
/* Backward of gsplat's rasterization() over the visible entries, C = 1 (models/trainers/base.py:393-408: the trainer passes ACTIVATED
 * scales / opacities and post-activation colours [N,3]): what bds_project_view_bwd_list does, with the gradients of the activated
 * scales and opacities returned as they are and the colour gradient (record channels 0-2) scattered to v_colors [N,3] (may be
 * NULL).  Store mode: the caller zero-fills the dense arrays; rows of culled Gaussians stay zero. */
int bds_project_bwd_list(int64_t n_list, const int32_t *ids, const float *means, const float *quats, const float *scales,
                         const float *opacities, const float *viewmat, const float *K, int W, int H, float eps2d,
                         const float *v_records, float *v_means, float *v_quats, float *v_scales, float *v_opacities, float *v_colors,
                         float *v_viewmat_slots, float *grad2d, float *absgrad2d, bds_stream_t stream);
int bds_project_view_bwd_list(int64_t n_list, const int32_t *ids, const float *means, const float *quats, const float *scales,
                              const float *opacities, const float *viewmat, const float *K, int W, int H, float eps2d,
                              const float *v_records, float *v_means, float *v_quats, float *v_log_scales, float *v_logits,
                              float *v_viewmat_slots, float *grad2d, float *absgrad2d, const int32_t *row_map, int accumulate,
                              bds_stream_t stream);
/* Zero the rows ids[0..n_list) of the five per-Gaussian gradient arrays (v_sh is [N,K,3]). */
int bds_view_grads_clear_list(int64_t n_list, const int32_t *ids, int K, float *v_means, float *v_quats, float *v_log_scales,
                              float *v_logits, float *v_sh, bds_stream_t stream);
/* v_*[ids[s]] += s_*[s]: compact rows (s_means [n_list,3] s_quats [n_list,4] s_log_scales [n_list,3] s_logits [n_list]
 * s_sh [n_list,K,3], e.g. a reduced exchange buffer) added to the dense arrays; entries with ids[s] < 0 are skipped. */
int bds_view_grads_add_list(int64_t n_list, const int32_t *ids, int K, const float *s_means, const float *s_quats,
                            const float *s_log_scales, const float *s_logits, const float *s_sh, float *v_means, float *v_quats,
                            float *v_log_scales, float *v_logits, float *v_sh, bds_stream_t stream);

/* The grids of ONE image picked by an image index that lives in DEVICE memory: a captured (hipGraph) view whose image changes from
 * replay to replay -- the reference draws a random image every step (tools/train.py:257) and indexes the grid parameters with it
 * (models/modules.py:507-512, `int(image_infos["img_idx"][0][0])`, a host read-back there).  levels[l]: grid / v_grid = the FULL
 * parameter [n_img,12,gl,gy,gx] and its gradient, n_avg = n_img.  select: sel[l] [12,gl,gy,gx] = grid_l[*img_idx_dev];
 * select_bwd: v_grid_l[*img_idx_dev] += v_sel[l], then v_sel[l] = 0.  An index outside [0, n_img) selects / adds nothing (the
 * staging grids keep their contents; select_bwd still clears v_sel) and sets *error_pinned = 1 (page-locked int32, sticky, may be
 * NULL): the host cannot see a device-side index otherwise. */
int bds_bilagrid_select(int nlevels, const bds_bilagrid_level_t *levels, const int32_t *img_idx_dev, float *const *sel,
                        int32_t *error_pinned, bds_stream_t stream);
int bds_bilagrid_select_bwd(int nlevels, const bds_bilagrid_level_t *levels, const int32_t *img_idx_dev, float *const *v_sel,
                            int32_t *error_pinned, bds_stream_t stream);

/* The colour transform's backward WITHOUT its last stage, and the compositor's backward that finishes it (one camera, RGB+ED; the
 * fused view, models/trainers/base.py:393-419 + scene_graph.py:86-120,292-294 as one backward).  The last stage of
 * bds_bilagrid_ms_ed_bwd -- guidance route added to the direct route, clamp(max=1) / sky blend / expected-depth backward -- is a
 * per-pixel function of arrays that exist by then; bds_rasterize_bwd_ms evaluates it for its tile's pixels while it waits for the
 * tile's first records, so v_render [H,W,4] and v_alpha [H,W] are never written and read back and one pass over the image (45 us,
 * 172 MB at 1080p) disappears.  _deferrable: 1 when every level has one grid (gl <= 8) and a factor that is 1 or a power of two
 * dividing H and W (and bit 2 of bds_set_option(7, ..) is clear), else 0 -- use bds_bilagrid_ms_ed_bwd then.  _deferred: v_direct
 * [H,W,4] receives the direct-route gradient (channels 0-2); the grids' gradients are complete on return.  bds_rasterize_bwd_ms:
 * bds_rasterize_bwd / _dev (M_dev NULL: M_capacity is the host-side count) for C = 1, CH = 4, no backgrounds, with the image
 * gradient formed from (levels, ms_ws: the transform's workspace), render [H,W,4] (the compositor's forward output), sky,
 * v_depth / v_alpha_in (may be NULL) and v_direct; writes v_sky [H,W,3] (may be NULL). */
int bds_bilagrid_ms_ed_bwd_deferrable(int nlevels, const bds_bilagrid_level_t *levels, int H, int W);
int bds_bilagrid_ms_ed_bwd_deferred(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render,
                                    const float *alpha, const float *sky, void *ws, size_t ws_bytes, const float *v_rgb_out,
                                    float *v_direct, bds_stream_t stream);
int bds_rasterize_bwd_ms(int64_t n_records, int64_t M_capacity, const uint64_t *M_dev, const float *records, int W, int H,
                         int tile_size, int list_tile_size, int tile_w, int tile_h, const int32_t *isect_offsets,
                         const int32_t *flatten, const float *alphas, const float *t_final, const int32_t *last_ids, float *v_records,
                         int absgrad, const int32_t *tile_order, int nlevels, const bds_bilagrid_level_t *levels, void *ms_ws, size_t ms_ws_bytes,
                         const float *render, const float *sky, const float *v_depth, const float *v_alpha_in, const float *v_direct,
                         float *v_sky, bds_stream_t stream);

/* Names (as rocprofv3 prints them, without "bds::" and the argument list; comma-separated, launch order) of the kernels the bilateral
 * transform of this configuration launches under the current options (bds_set_option(7, ..)): forward (train != 0: with the L1 / TV
 * loss on the launch) or backward.  Measurement plumbing for bench.py's counter look-up; no reference counterpart. */
int bds_bilagrid_kernel_names(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, int backward, int train, char *buf,
                              int buf_len);

/* Name (as rocprofv3 prints it, without the "bds::" prefix and the argument list) of the compositor kernel that a launch with
 * these switches runs (backward: 0 = forward, 1 = backward, 2 = backward with the colour transform's deferred epilogue,
 * bds_rasterize_bwd_ms); measurement plumbing for bench.py's counter look-up. */
int bds_rasterize_kernel_name(int backward, int CH, int absgrad, int list_tile_size, char *buf, int buf_len);

/* ---- device-count forms: one view without a host read-back (capturable in a hipGraph) -----------------------------------------
 * gsplat's rasterization() reads the intersection count back to size its lists (one host wait per view at
 * models/trainers/base.py:393-408).  These forms take CAPACITIES from the host (what the previous visit of the camera needed, plus
 * head-room) and the actual counts from device memory: the first words of the prepare workspace hold, as uint64,
 * {M, visible entries, M effective, visible effective, overflow} (byte offsets: bds_isect_counts_offset(0..4)).  The effective
 * counts are what every later stage sizes itself by; both are ZERO when a count outgrew its capacity -- the view then renders
 * nothing instead of overrunning a buffer, and the host, which looks at `counts_pinned` (page-locked int64[3] = M, visible,
 * overflow; written by the GPU; may be NULL) whenever it likes, provisions more and repeats the view.  Launches are sized by the
 * capacities; surplus workgroups see no elements.  Packed lists only (n_visible_capacity <= 2^(32 - bits(C*tiles))), else
 * BDS_ECAPACITY.  Lists, offsets and images are bit-identical to the host-count forms.  `compact`: bit 0 as in bds_isect_prepare,
 * bit 1 = the workspace already holds the visible counts and cleared tables of bds_project_view_prepare_fwd (one launch less). */
size_t bds_isect_counts_offset(int which);
int bds_isect_prepare_dev(int C, int64_t N, const float *means2d, const int32_t *radii, const float *depths, const float *conics,
                          const float *opacities, int tile_size, int tile_w, int tile_h, int32_t *tiles_per_gauss, void *ws,
                          size_t ws_bytes, int64_t M_capacity, int64_t n_visible_capacity, int64_t *counts_pinned, int compact,
                          bds_stream_t stream);
/* ws2: bds_isect_build_workspace_bytes(C, N, M_capacity); flatten_ids [M_capacity] */
int bds_isect_build_dev(int C, int64_t N, int64_t M_capacity, int64_t n_visible_capacity, const float *means2d, const int32_t *radii,
                        const float *depths, const float *conics, const float *opacities, int tile_size, int tile_w, int tile_h,
                        const void *ws, size_t ws_bytes, void *ws2, size_t ws2_bytes, int32_t *flatten_ids, int32_t *isect_offsets,
                        int compact, bds_stream_t stream);
/* bds_splat_pack with the record count on the device (n_dev -> visible effective).  Optionally clears, on the way, the gradient
 * record of every packed row (zero_records [n_capacity, BDS_GRAD_RECORD_FLOATS]: what bds_rasterize_bwd accumulates into) and a
 * tail of zero_tail_floats (multiple of 4) floats (the camera-pose gradient slots): no fill launches of their own.  schedule
 * (optional): the schedule buffer the following bds_rasterize_fwd_dev fills in its binned form -- its header is cleared here too. */
int bds_splat_pack_dev(int64_t n_capacity, const uint64_t *n_dev, int CH, const int32_t *ids, const float *means2d, const float *conics,
                       const float *colors, const float *opacities, const int32_t *radii, float *records, float *zero_records,
                       float *zero_tail, int64_t zero_tail_floats, int32_t *schedule, bds_stream_t stream);
/* bds_splat_pack_sh with the record count on the device and the clearing options of bds_splat_pack_dev: the SH colours of the visible
 * Gaussians are evaluated by the pack itself (no pass over all N, no dense colour arrays); sh_rgb [n_capacity, 3] in list order. */
int bds_splat_pack_sh_dev(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, int K, int degrees_to_use, const float *means,
                          const float *cam_pos, const float *coeffs, const float *means2d, const float *conics, const float *depths,
                          const float *opacities, const int32_t *radii, float *records, float *sh_rgb, float *zero_records,
                          float *zero_tail, int64_t zero_tail_floats, int32_t *schedule, bds_stream_t stream);
/* bds_rasterize_fwd / _bwd with the list length on the device (M_dev -> M effective) */
/* (tile_order, optional: the schedule buffer of bds_rasterize_bwd_schedule.  Every compositing wave knows how far into its list its
 * tile blended when it ends, and leaves the backward's schedule itself: BINNED form (bds_set_option(8, 1), default) -- one atomic
 * drops the tile into the bin of its length (32 bins a factor 2^(1/4) apart, longest first) of its XCD's range of tiles, the
 * backward's workgroups find their tile by a prefix walk over the 32 counts, NO launch between the passes; the header must be clear when the
 * forward starts (the record pack's `schedule` argument).  bds_set_option(8, 0): the waves leave their keys and
 * bds_rasterize_bwd_schedule_sort -- one launch, a no-op in the binned form -- writes the sorted schedule.)
 * split_len (> 0: one camera, four channels, lists of tiles larger than 16 px, the binned schedule, i.e. the fused view; 0 = off): a
 * tile whose list-tile list holds >= split_len entries is composited by FOUR waves, one 16 x 4 strip each (one pixel per lane; the
 * candidates filtered per strip), instead of one: a one-workgroup kernel lists those tiles behind the schedule words (at most
 * split_cap of them; a long tile beyond that is taken by one wave like any other) and the launch is [4 x split_cap strip workgroups,
 * first | one workgroup per tile].  For views in which a few tiles collect thousands of small splats (the vanishing point of a
 * street: the launch waits for those waves).  Same pixels in the same order: images bit-identical, gradients to the order of their
 * atomics.  Pass the SAME values and the same tile_order buffer to the backward.  No reference counterpart (gsplat runs 256 threads
 * per tile everywhere).
 * split_pool (> 0, with split_len; 0 = off): int32 words of a pool behind the schedule words -- the tile_order buffer then holds
 * bds_rasterize_schedule_ints(..) + bds_rasterize_split_pool_ints(.., split_cap, split_pool, M_capacity) words.  The long tiles of a
 * list tile share one walk of its list (per entry the mask of the sub-tiles it reaches), then one workgroup per long tile leaves the
 * tile's own candidates in the pool, in list order; the tile's strips (forward and
 * backward) walk that instead of the whole list-tile list (the front camera of a lidar-initialised street: 36 k entries per strip
 * wave -> the ~8 k that reach the tile).  A tile the pool has no room for keeps the list-tile list: never an error.  last_ids of a
 * refined tile are positions in the pool. */
int bds_rasterize_fwd_dev(int C, int64_t n_records, int64_t M_capacity, const uint64_t *M_dev, int CH, const float *records,
                          const float *backgrounds, int W, int H, int tile_size, int list_tile_size, int tile_w, int tile_h,
                          const int32_t *isect_offsets, const int32_t *flatten, float *render, float *alphas, float *t_final,
                          int32_t *last_ids, int32_t *tile_order, int split_len, int split_cap, int64_t split_pool, bds_stream_t stream);
int64_t bds_rasterize_split_pool_ints(int C, int tile_w, int tile_h, int split_cap, int64_t split_pool, int64_t M_capacity);
int bds_rasterize_bwd_schedule_sort(int C, int tile_w, int tile_h, int32_t *tile_order, bds_stream_t stream);
int bds_rasterize_bwd_dev(int C, int64_t n_records, int64_t M_capacity, const uint64_t *M_dev, int CH, const float *records,
                          const float *backgrounds, int W, int H, int tile_size, int list_tile_size, int tile_w, int tile_h,
                          const int32_t *isect_offsets, const int32_t *flatten, const float *alphas, const float *t_final,
                          const int32_t *last_ids, const float *v_render, const float *v_alphas, float *v_records, int absgrad,
                          const int32_t *tile_order, int split_len, int split_cap, int64_t split_pool, bds_stream_t stream);
/* the list-driven backward kernels and the row-wise clear with the list length on the device (n_dev -> visible effective) */
int bds_sh_view_bwd_list_dev(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, int K, int degrees_to_use,
                             const float *means, const float *cam_pos, const float *sh_rgb, int sh_rgb_by_rank,
                             const float *v_records, float *v_coeffs, const int32_t *row_map, int accumulate, bds_stream_t stream);
int bds_project_view_bwd_list_dev(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, const float *means,
                                  const float *quats, const float *scales, const float *opacities, const float *viewmat,
                                  const float *K, int W, int H, float eps2d, const float *v_records, float *v_means, float *v_quats,
                                  float *v_log_scales, float *v_logits, float *v_viewmat_slots, float *grad2d, float *absgrad2d,
                                  const int32_t *row_map, int accumulate, bds_stream_t stream);
/* (grad2d / absgrad2d [N,2], optional: the same rows of a view's PERSISTENT screen-space gradient arrays are cleared as well -- the
 * list-driven projection backward stores the visible rows, so a buffer cleared by the previous visit's list needs no dense fill.
 * The five parameter-gradient pointers may ALL be NULL: only the screen-space arrays are cleared then -- a loop whose optimizer
 * clears the gradients as it consumes them, bds_adam_step_consume) */
int bds_view_grads_clear_list_dev(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, int K, float *v_means,
                                  float *v_quats, float *v_log_scales, float *v_logits, float *v_sh, float *grad2d, float *absgrad2d,
                                  bds_stream_t stream);

/* ---- multi-GPU exchange of the visible rows (no reference counterpart; dist.FrameExchange) ---------------------------------------
 * mask [N] uint8: the element-wise OR over the ranks of "this rank's view sees Gaussian g" (radii > 0).  In two launches:
 * row_map [N] i32 = slot of every Gaussian in the union (cumsum(mask) - 1, clamped to [0, capacity - 1]); ids [capacity] i32 =
 * Gaussian at slot s, -1 beyond the union; rows [0, count) of the five compact sub-arrays (b_means [capacity,3] b_quats [capacity,4]
 * b_log_scales [capacity,3] b_logits [capacity] b_sh [capacity,K,3]) zeroed; count -> *count_dev (may be NULL) and count_pinned
 * (page-locked int64, written by the GPU; may be NULL).  A union larger than the capacity is reported through the count (rows
 * beyond it collide in the last slot: the caller repeats the frame with larger buffers).  ws: bds_union_slots_workspace_bytes(N). */
size_t bds_union_slots_workspace_bytes(int64_t N);
int bds_union_slots(int64_t N, const uint8_t *mask, int64_t capacity, int K, int32_t *row_map, int32_t *ids, float *b_means,
                    float *b_quats, float *b_log_scales, float *b_logits, float *b_sh, void *ws, size_t ws_bytes,
                    uint64_t *count_dev, int64_t *count_pinned, bds_stream_t stream);

/* RGB+ED form of the fused image transform: the input is the compositor's 4-channel render [H*W,4] (RGB +
 * accumulated depth, gsplat render_mode "RGB+ED") and its alpha.  Forward additionally writes the expected depth
 * depth [H*W] = render[.,3] / max(alpha, 1e-10) (the normalise inside gsplat's rasterization(); split as in
 * models/trainers/base.py:414-419).  Backward returns v_render [H*W,4] and the TOTAL alpha gradient: colour
 * transform (sky blend) + expected depth + the caller's own v_opacity; v_depth / v_opacity / v_sky may be NULL.
 * Same workspace as bds_bilagrid_ms_fwd/bwd. */
int bds_bilagrid_ms_ed_fwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render,
                           const float *alpha, const float *sky, void *ws, size_t ws_bytes, float *rgb_out,
                           float *depth_out, bds_stream_t stream);
/* bds_bilagrid_ms_ed_fwd with the training loss of the direct step riding on the same launch (no further pass over the image): the
 * full-resolution kernel also compares the pixel it just produced with target [H*W,3] -- loss_out (slotted as in bds_l1_tv_train) +=
 * mean|rgb_out - target|, v_rgb_out [H*W,3] = sign(rgb_out - target) * v_loss / (3 H W) -- and tv_nlevels extra levels' worth of
 * workgroups add sum_l tv_weights[l] * TV(tv_levels[l].grid) to loss_out and v_loss * its gradient to tv_levels[l].v_grid (atomics;
 * may be NULL).  Same result as bds_bilagrid_ms_ed_fwd followed by bds_l1_tv_train.  (models/trainers/base.py:518-565 rgb term +
 * `losses.affine`, models/modules.py:445,466-472.) */
int bds_bilagrid_ms_ed_train_fwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render, const float *alpha,
                                 const float *sky, void *ws, size_t ws_bytes, float *rgb_out, float *depth_out, const float *target,
                                 int tv_nlevels, const bds_bilagrid_level_t *tv_levels, const float *tv_weights, float v_loss,
                                 float *loss_out, int loss_slots, float *v_rgb_out, bds_stream_t stream);
int bds_bilagrid_ms_ed_bwd(int nlevels, const bds_bilagrid_level_t *levels, int H, int W, const float *render,
                           const float *alpha, const float *sky, void *ws, size_t ws_bytes, const float *v_rgb_out,
                           const float *v_depth, const float *v_opacity, float *v_render, float *v_alpha, float *v_sky,
                           bds_stream_t stream);

/* ---- photometric L1 (the step right after the path) ---------------------------------
 * models/trainers/base.py:518-529: mean |a - b| over n floats.  out [1] is ACCUMULATED (caller zero-fills, so that
 * the TV terms of bds_bilagrid_tv_fwd can land in the same scalar); a, b 16-byte aligned.
 * bwd: v_a = sign(a - b) * v_out / n with v_out a device scalar. */
int bds_l1_mean_fwd(int64_t n, const float *a, const float *b, float *out, bds_stream_t stream);
int bds_l1_mean_bwd(int64_t n, const float *a, const float *b, const float *v_out, float *v_a, bds_stream_t stream);

/* SSIM term of the image loss (models/trainers/base.py:114,541: pytorch_msssim.SSIM(data_range=1, size_average=True,
 * channel=3), loss = 1 - ssim(gt, pred)): 11x11 Gaussian window (sigma 1.5), valid region, K = (0.01, 0.03).
 * target / pred [H,W,CH] (H, W >= 11).  fwd ACCUMULATES the mean SSIM into ssim_out [1] (caller zero-fills) and, when
 * ws != NULL (bds_ssim_workspace_bytes), stores the per-pixel derivatives the backward needs; bwd writes
 * v_pred = v_ssim * d(mean SSIM)/d pred with v_ssim a device scalar (pass -upstream for the 1 - ssim loss). */
size_t bds_ssim_workspace_bytes(int H, int W, int CH);
int bds_ssim_fwd(int H, int W, int CH, const float *target, const float *pred, float *ssim_out, void *ws, size_t ws_bytes,
                 bds_stream_t stream);
int bds_ssim_bwd(int H, int W, int CH, const float *target, const float *pred, const void *ws, size_t ws_bytes,
                 const float *v_ssim, float *v_pred, bds_stream_t stream);

/* Per-pixel terms of the reference's image loss in one pass each way (models/trainers/base.py:518-565; loss functions
 * :230-250 = models/losses.py binary_cross_entropy and DepthLoss(normalize=False, use_inverse_depth=False)):
 *   terms[0] = w_rgb   * mean |pixels*valid - rgb*valid|                          rgb, pixels [P,3]
 *   terms[1] = w_mask  * mean BCE(opacity*valid, (1 - sky_masks)*valid)           opacity, sky_masks [P] (both or neither)
 *   terms[2] = w_depth * mean over valid lidar returns of |.| (or (.)^2 if depth_l2) of depth*hit - lidar*hit,
 *              hit = (lidar > 0)*valid, valid returns: 0.01 < gt < max_depth, pred > 1e-4   depth, lidar [P] (both or neither)
 * valid = 1 - egocar [P] (NULL: 1).  sums [4] is scratch that the backward reads (three numerators + the depth count);
 * v_terms [3] are the upstream gradients of the three terms (device); v_opacity / v_depth may be NULL. */
int bds_pixel_loss_fwd(int64_t P, const float *rgb, const float *pixels, const float *opacity, const float *sky_masks,
                       const float *depth, const float *lidar, const float *egocar, float w_rgb, float w_mask, float w_depth,
                       int depth_l2, float max_depth, float *sums, float *terms, bds_stream_t stream);
int bds_pixel_loss_bwd(int64_t P, const float *rgb, const float *pixels, const float *opacity, const float *sky_masks,
                       const float *depth, const float *lidar, const float *egocar, float w_rgb, float w_mask, float w_depth,
                       int depth_l2, float max_depth, const float *sums, const float *v_terms, float *v_rgb, float *v_opacity,
                       float *v_depth, bds_stream_t stream);

/* ---- Adam step on one parameter tensor (first slice) --------------------------------------
 * torch.optim.Adam as the reference trainer configures it (models/trainers/base.py:201-222: per-group lr / eps /
 * weight_decay, betas (0.9, 0.999), amsgrad off), one streaming pass, in place on param / exp_avg / exp_avg_sq.
 * `step` is the 1-based step count AFTER the increment (torch's state["step"]); the hyper-parameters are doubles (Python
 * floats): 1 - beta and the bias corrections are formed in double and rounded to fp32 once, as torch does. */
int bds_adam_step(int64_t n, float *param, const float *grad, float *exp_avg, float *exp_avg_sq, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t step, bds_stream_t stream);
/* The same, and the gradient is cleared as it is read ("consume and clear"): a training loop whose backward ACCUMULATES into
 * persistent gradient buffers (dist.FlatGradients / graph_view.FrameGraph) then needs no clearing pass before the next step --
 * what optimizer.zero_grad() (tools/train.py:266) costs the reference as one more pass over every gradient. */
int bds_adam_step_consume(int64_t n, float *param, float *grad, float *exp_avg, float *exp_avg_sq, double lr, double beta1,
                          double beta2, double eps, double weight_decay, int64_t step, bds_stream_t stream);
/* The same update for a parameter [n_rows, width] whose gradient is a column range of a wider row block: element (r, c) at
 * grad[r * grad_stride + c] (the [N,16] row form of the four small per-Gaussian gradients, see bds_project_view_bwd_list);
 * consume != 0 clears the gradient as it is read. */
int bds_adam_step_rows(int64_t n_rows, int width, int64_t grad_stride, float *param, float *grad, float *exp_avg, float *exp_avg_sq,
                       double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, int consume,
                       bds_stream_t stream);

/* The same update for up to four parameter tensors [N, widths[t]] whose gradients are the column ranges [col0[t], col0[t] + widths[t])
 * of ONE [N,16] row block (64-byte rows; the four small per-Gaussian gradients, see bds_project_view_bwd_list), in one launch that
 * reads -- and, consuming, clears -- every row once: stepping the four tensors one after the other reads every 64-byte row four times. */
int bds_adam_step_rowblock(int64_t N, float *grad_block, int n_parts, float *const *params, float *const *exp_avgs,
                           float *const *exp_avg_sqs, const int *col0, const int *widths, const double *lrs, const double *beta1s,
                           const double *beta2s, const double *eps, const double *weight_decays, const int64_t *steps, int consume,
                           bds_stream_t stream);
/* bds_adam_step / _consume / _rows for up to 12 tensors in ONE launch (the trainer's ~10 small groups, models/trainers/base.py:201-226:
 * one launch per tensor is mostly launch gap).  Arrays of n_tensors entries; widths[t] = 0: a contiguous gradient, else element
 * (r, c) of tensor t's gradient at grads[t][r * grad_strides[t] + c] with c < widths[t]; steps[t]: the tensor's own 1-based step.
 * The same arithmetic per element: bit-equal to the single-tensor passes. */
int bds_adam_step_multi(int n_tensors, float *const *params, float *const *grads, float *const *exp_avgs, float *const *exp_avg_sqs,
                        const int64_t *counts, const int *widths, const int64_t *grad_strides, const double *lrs, const double *beta1s,
                        const double *beta2s, const double *eps, const double *weight_decays, const int64_t *steps, int consume,
                        bds_stream_t stream);
/* Deferred ("row-lazy") Adam for a parameter [N, row_floats] (row_floats <= 256) that only the rows of a view's visible-id list read
 * -- the SH coefficients -- with the numbers of the dense pass, bit for bit: the reference steps one dense Adam after every
 * single-view iteration (tools/train.py:252-283, models/trainers/base.py:222-226,502-516) and a row whose gradient is zero still
 * moves, so the steps a row missed are REPLAYED (same fp32 operations, same order, per-step scalars from `table`) when the row is
 * next needed.  last_step [N] i32: the step each row is current for; table [table_steps][4] floats (16-byte aligned): per-step
 * (lr_s/(1-b1^s) for columns < split_col, the same for columns >= split_col with lr_b, sqrt(1-b2^s), -), slot s % table_steps
 * written by the with_step launch of step s.  One call = "bring the rows of the list (ids [n_capacity] i32, entries < 0 skipped;
 * NULL: rows 0 .. n_capacity-1; n_dev: optional device-side count) to step t":
 *   with_step = 0: t = *clock_dev if clock_dev else `step`; rows with last < t replay the zero-gradient steps last+1 .. t.  Enqueued
 *                  inside a view's forward (after the visible list, before the record pack) with clock_dev, so that a captured
 *                  graph follows the optimizer; or densely before anything else reads the tensor / every table_steps-1 steps.
 *   with_step = 1: t = `step` (1-based, after the increment); rows with last < t replay last+1 .. t-1 and take step t from grad
 *                  [N, row_floats] (cleared as read when consume); rows already at t are skipped (a row listed by two views of a
 *                  frame steps once); table[t % table_steps] is written and *clock_dev = t.
 * The caller guarantees t - last < table_steps for every row it lists.  No reference counterpart (torch.optim.Adam is dense). */
int bds_adam_rows_advance(int64_t n_capacity, const uint64_t *n_dev, const int32_t *ids, int64_t N, int row_floats, int split_col,
                          float *param, float *grad, int consume, float *exp_avg, float *exp_avg_sq, int32_t *last_step,
                          int32_t *clock_dev, int64_t step, int with_step, float *table, int table_steps, double lr_a, double lr_b,
                          double beta1, double beta2, double eps, double weight_decay, bds_stream_t stream);

/* Per-step densification statistics of one set of Gaussians in one launch (models/trainers/base.py:279-297 +
 * models/gaussians/vanilla.py:163-191): grad2d [N,2] is info["means2d"].absgrad (or .grad) BEFORE the trainer's
 * width/2, height/2, batch_size scaling; radii [N] i32.  first != 0 reproduces the reference's initialising call
 * (xys_grad_norm = norm for every Gaussian, vis_counts = 1 for every Gaussian, max_2Dsize from zero). */
int bds_densify_stats(int64_t N, const float *grad2d, const int32_t *radii, int width, int height, int batch_size,
                      int last_size, int first, float *xys_grad_norm, float *vis_counts, float *max_2Dsize,
                      bds_stream_t stream);

/* ---- Adaptive density control: split / duplicate / cull + Adam-state surgery (second slice) -----
 * VanillaGaussians.refinement_after (models/gaussians/vanilla.py:205-304) with split_gaussians (:336-363), dup_gaussians
 * (:365-376), cull_gaussians (:306-334) and dup_in_optim / remove_from_optim (models/gaussians/basics.py:162-206), done as
 * ONE plan over the Gaussians followed by one write of every array straight into its final (culled) layout
 *   [ kept originals | kept split children, sample-major | kept dup children ]      (the order the reference's cat + cull gives).
 *
 * bds_refine_plan: flags [N] u8 (bit 0 split, 1 dup, 2 keep original, 3 keep its split children, 4 keep its dup child),
 * ranks [N,4] u32 (16-byte aligned; exclusive ranks among: split parents, kept originals, kept split parents, kept dup
 * parents) and totals [5] i64 on the device = {n_split, n_dup, kept originals KO, kept split parents KS, kept dup KD};
 * the new size is KO + samps*KS + KD and the host needs n_split for the noise tensor ([samps*n_split, 3], the reference's
 * torch.randn at vanilla.py:343).  The host decides the step-dependent switches exactly as the reference does:
 *   do_densify      step < stop_split_at and step % reset_interval > max(num_train_images, refine_interval)   (:212-215)
 *   size_thresh     densify_size_thresh * scene_scale;   split_by_screen: step < stop_screen_size_at            (:224-230)
 *   do_cull         step % reset_interval > max(num_train_images, refine_interval)                               (:279)
 *   cull_by_scale   step > reset_alpha_interval, cull_scale_thresh = cull_scale_thresh * scene_scale             (:314-320)
 *   cull_by_screen  additionally step < stop_screen_size_at                                                      (:321-324)
 * xys_grad_norm / vis_counts may be NULL when do_densify == 0; max_2Dsize may be NULL (treated as zeros).
 * extra_cull [N] u8 (may be NULL): a caller-computed cull mask over the INPUT rows, OR-ed into the decision for the originals when
 * do_cull != 0 (children are unaffected).  It serves the node classes' box test (models/nodes/rigid.py:302-303), which the
 * reference evaluates after the children were appended: the host runs a first plan for split / dup / the common culls, then
 * bds_refine_out_of_bound on the NEW means and a second, cull-only plan with that mask -- two order-preserving compactions leave
 * the rows, and their order, of the reference's single `culls` mask.
 * temp: bds_refine_plan_temp_bytes(N) bytes of scratch.  N < 2^31. */
size_t bds_refine_plan_temp_bytes(int64_t N);
int bds_refine_plan(int64_t N, const float *xys_grad_norm, const float *vis_counts, const float *max_2Dsize,
                    const float *log_scales, const float *logits, const uint8_t *extra_cull, int do_densify, float grad_thresh,
                    float size_thresh, int split_by_screen, float split_screen_size, int do_cull, float cull_alpha_thresh,
                    int cull_by_scale, float cull_scale_thresh, int cull_by_screen, float cull_screen_size, uint8_t *flags,
                    uint32_t *ranks, int64_t *totals, void *temp, size_t temp_bytes, bds_stream_t stream);

/* RigidNodes.get_out_of_bound_mask (models/nodes/rigid.py:374-383; DeformableNodes inherits it): mask[g] = 1 when
 * |means[g, i]| > instances_size[point_ids[g], i] / 2 for any axis i (means in the object frame, instances_size [n_instances,3],
 * point_ids [N] i64 -- the reference's [N,1] column).  An id outside [0, n_instances) is reported out of bound. */
int bds_refine_out_of_bound(int64_t N, const float *means, const int64_t *point_ids, int64_t n_instances,
                            const float *instances_size, uint8_t *mask, bds_stream_t stream);

/* New means [N',3] and log-scales [N',3]: split children are mean + R(q/|q|) (exp(log_scale) * noise) (vanilla.py:343-348);
 * a split parent and its children get log(exp(log_scale) / 1.6) (:356-359); dup children copy the (possibly shrunk) parent. */
int bds_refine_geometry(int64_t N, int samps, const uint8_t *flags, const uint32_t *ranks, const int64_t *totals,
                        const float *samples, const float *means, const float *quats, const float *log_scales,
                        float *new_means, float *new_log_scales, bds_stream_t stream);

/* Any other per-Gaussian array src [N,width] -> dst [N',width]: children receive the parent's row (zero_children = 0:
 * features, opacities, quaternions) or zeros (zero_children = 1: exp_avg / exp_avg_sq, basics.py:191-201). */
int bds_refine_rows(int64_t N, int width, int samps, const uint8_t *flags, const uint32_t *ranks, const int64_t *totals,
                    const float *src, float *dst, int zero_children, bds_stream_t stream);

/* Opacity reset (vanilla.py:286-299): logit = logit(min(sigmoid(logit), reset_value)) in place; the opacity group's
 * exp_avg / exp_avg_sq (either may be NULL) are zeroed. */
int bds_opacity_reset(int64_t N, float *logits, float reset_value, float *exp_avg, float *exp_avg_sq, bds_stream_t stream);

/* Regularisers of the reference's image loss (models/trainers/base.py:566-585, 638-659), one pass each way:
 *   terms[0] opacity entropy       (-p log p).mean(), p = clamp(opacity, 1e-6, 1-1e-6)                       (opacity != NULL)
 *   terms[1] inverse-depth smoothness  kornia.losses.inverse_depth_smoothness_loss(1/(depth+1e-5), pixels)    (depth != NULL;
 *            kornia is an external package absent here: restated from its published definition, PARITY UNPINNED)
 *   terms[2] dynamic-region L1     |pixels*valid - rgb*valid| averaged over {dyn_opacity > dyn_threshold and valid} (dyn_opacity != NULL;
 *            0 when the mask is empty, where the reference adds no term)
 * opacity / depth / dyn_opacity / egocar [H*W], pixels / rgb [H*W,3]; sums [5] and terms [3] on the device.  The backward writes
 * (not accumulates) v_opacity / v_depth [H*W], v_rgb [H*W,3] (each may be NULL) for the UNWEIGHTED terms scaled by v_terms [3]. */
int bds_reg_loss_fwd(int H, int W, const float *opacity, const float *depth, const float *pixels, const float *rgb,
                     const float *dyn_opacity, const float *egocar, float dyn_threshold, float *sums, float *terms,
                     bds_stream_t stream);
int bds_reg_loss_bwd(int H, int W, const float *opacity, const float *depth, const float *pixels, const float *rgb,
                     const float *dyn_opacity, const float *egocar, float dyn_threshold, const float *sums, const float *v_terms,
                     float *v_opacity, float *v_depth, float *v_rgb, bds_stream_t stream);

/* ---- Cube-map sky (the ROCm replacement of nvdiffrast's cube texture) -----------------------------
 * EnvLight.forward (models/modules.py:176-211): out[i] = bilinear cube-map lookup of tex [6,res,res,channels] along
 * dirs[i] @ rot^T (rot: 9 floats on the device, row-major, NULL = identity; the reference's to_opengl, :189,196) --
 * dr.texture(tex[None], l, filter_mode='linear', boundary_mode='cube') (:202).  OpenGL face order / orientation; taps that
 * fall off a face come from the neighbouring face; a non-finite direction yields zeros.  PARITY UNPINNED (nvdiffrast is
 * absent and unpinned, README.md:83): see csrc/envlight.hip.
 * bwd ACCUMULATES into v_tex (caller zeroes it); directions carry no gradient (the reference's viewdirs are data).
 * width > 0 tells bwd that dirs is a row-major image [n / width, width, 3] (n % width == 0): 16x16 pixel tiles then pre-sum
 * their taps in LDS before touching v_tex (channels == 3 only); width = 0: one global atomic per tap and channel. */
int bds_cubemap_fwd(int64_t n, int res, int channels, const float *dirs, const float *rot, const float *tex, float *out,
                    bds_stream_t stream);
int bds_cubemap_bwd(int64_t n, int res, int channels, int width, const float *dirs, const float *rot, const float *v_out,
                    float *v_tex, bds_stream_t stream);

/* ---- Colour-correct post-process of the evaluation path ----------------------------------------
 * One iteration of lib_bilagrid.color_correct (bilateral/lib_bilagrid.py:56-120; called per frame at
 * models/video_utils_color_correction.py:201) as one streaming pass:
 *   cur = warp ? clip(expand(cur_in) @ warp, 0, 1) : cur_in          expand(r,g,b) = [rr, rg, rb, gg, gb, bb, r, g, b, 1] (:98-104)
 *   mask0 [P] u8: bit c = channel c of the ORIGINAL image is in [eps, 1-eps] -- written when warp == NULL (first pass), read otherwise
 *   cur_out [P,3] (may be NULL) receives cur
 *   acc [3,65] doubles (may be NULL; ACCUMULATED into, caller zeroes): per output channel c, over the pixels with mask0 bit c set and
 *   cur_c, ref_c in [eps, 1-eps] (:110): the 55 upper-triangle entries (row-major) of sum expand expand^T, then the 10 entries of
 *   sum expand * ref_c.  The caller solves the three 10x10 systems (float64) -> next warp [10,3] row-major float. */
int bds_color_correct_step(int64_t P, const float *cur_in, const float *ref, const float *warp, float eps, uint8_t *mask0,
                           float *cur_out, double *acc, bds_stream_t stream);

/* ---- Fused head of the neural bilateral variants ---------------------------------------------------
 * NeuralBilateralAffineTransform / MultiScaleNeuralBilateralAffineTransform (models/modules.py:595-820): per pixel the sliced
 * features [P,F] go through `affine_network` = Linear(F,hidden) tanh Linear(hidden,hidden) tanh Linear(hidden,12), all without bias
 * (:621-627, :700-706; weights in torch's [out, in] layout: w1 [hidden,F], w2 [hidden,hidden], w3 [12,hidden]), and the 12 outputs
 * are the row-major 3x4 map the trainer applies with a residual (models/trainers/scene_graph.py:99-106):
 *   affine [P,12] (may be NULL)  = the network's output (what the modules' forward returns, reshaped [1,H,W,3,4] by the caller)
 *   out    [P,3]  (may be NULL)  = A[:, :3] rgb + A[:, 3] (+ rgb when residual != 0)
 * One kernel each way on the FP32 matrix cores (exact f32 products and sums, csrc/mlp_head.hip); hidden must be 64 (the shipped
 * configs, configs/omnire_neuralbilateral.yaml:251-252, omnire_ms_neuralbilateral.yaml:249-250) and F one of 8, 16, 24, 32
 * (feature_dim x number of levels), anything else returns BDS_EINVAL.  feats / affine / v_feats / v_affine 16-byte aligned.
 * bwd takes the gradient of either output (v_out and / or v_affine, the other NULL) and writes v_feats [P,F] and v_rgb [P,3] (the
 * direct path through the application only: the path through the features belongs to the slice operator; needs v_out) -- each may be
 * NULL -- and the weight gradients v_w1 / v_w2 / v_w3 (each may be NULL; stored, or added to when accumulate_w != 0).  The hidden
 * activations are recomputed, nothing is kept from the forward.  temp: bds_mlp_head_bwd_temp_bytes(P, F) bytes (per-wave partial
 * weight gradients, summed in a fixed order: the result is deterministic). */
size_t bds_mlp_head_bwd_temp_bytes(int64_t P, int F);
int bds_mlp_head_fwd(int64_t P, int F, int hidden, const float *feats, const float *rgb, const float *w1, const float *w2,
                     const float *w3, int residual, float *out, float *affine, bds_stream_t stream);
int bds_mlp_head_bwd(int64_t P, int F, int hidden, const float *feats, const float *rgb, const float *w1, const float *w2,
                     const float *w3, int residual, const float *v_out, const float *v_affine, float *v_feats, float *v_rgb,
                     float *v_w1, float *v_w2, float *v_w3, int accumulate_w, void *temp, size_t temp_bytes, bds_stream_t stream);

/* The whole `transform` of the neural variants for one image, the feature slice folded in (the sliced features never reach
 * memory): NeuralBilateralAffineTransform.forward / MultiScaleNeuralBilateralAffineTransform.forward with guidance_factor = None
 * (models/modules.py:643-670, 728-790: one feature grid per level, sliced at the pixel grid linspace x linspace with the image's
 * gray value as guidance, levels concatenated) -> affine_network -> the trainer's application with the residual
 * (models/trainers/scene_graph.py:99-106).  levels[l].grid is ONE image's grid [nch, gl, gy, gx] (the test branch's mean over
 * neighbour grids is linear: the caller averages the grids); rgb / out / v_out / v_rgb [H,W,3].
 * bwd ACCUMULATES the grid gradients into levels[l].v_grid (same shape; caller zeroes; may be NULL), writes v_rgb (both routes: the
 * application and the guidance; may be NULL) and the weight gradients as bds_mlp_head_bwd does.
 * Built for: one level with gl = 8 and 8 / 16 / 24 / 32 features, or the two levels {gl 1, 8 features} + {gl 8, 8 features}
 * (configs/omnire_neuralbilateral.yaml:246-252, omnire_ms_neuralbilateral.yaml:247-250), plus gl = 4 with 24 features and
 * {gl 1, 8} + {gl 4, 8} (the shapes of the reference-generated golden vectors), any gx, gy with at most 64 cell
 * boundaries per axis, W <= 4096, hidden 64: bds_neural_image_ok() != 0; otherwise BDS_EINVAL and the two-step form
 * (bds_bilagrid_slice_feat_image_* + bds_mlp_head_*) applies.  temp: bds_neural_image_bwd_temp_bytes(F) bytes. */
typedef struct {
  const float *grid;
  float *v_grid;
  int gx, gy, gl, nch;
} bds_feat_level;
int bds_neural_image_ok(int H, int W, int n_levels, const bds_feat_level *levels, int hidden);
size_t bds_neural_image_bwd_temp_bytes(int F);
int bds_neural_image_fwd(int H, int W, int n_levels, const bds_feat_level *levels, int hidden, const float *rgb, const float *w1,
                         const float *w2, const float *w3, int residual, float *out, bds_stream_t stream);
int bds_neural_image_bwd(int H, int W, int n_levels, const bds_feat_level *levels, int hidden, const float *rgb, const float *w1,
                         const float *w2, const float *w3, int residual, const float *v_out, float *v_rgb, float *v_w1, float *v_w2,
                         float *v_w3, int accumulate_w, void *temp, size_t temp_bytes, bds_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BDS_H */
