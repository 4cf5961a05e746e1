/*
 * tissue_hip.h -- C-ABI of libtissue_hip.so (MI355X / gfx950 hot path).
 *
 * The reference (kasirershahartau/tissue_image_processing) is pure Python and has no
 * native boundary of its own; each entry point below replaces the third-party call
 * the reference makes at the cited line, so that the Python drop-in modules
 * (tissue_image_processing_amd/{basic_image_manipulations,surface_projection,
 * tissue_info,prediction_local}.py) can keep the reference's signatures and bind
 * these symbols through ctypes.
 *
 * Conventions
 *   - plain C types only; row-major contiguous arrays; explicit dims
 *   - return 0 on success, negative tip_status on error; text via tip_last_error()
 *   - `*_dev` variants take DEVICE pointers, run asynchronously on the calling
 *     thread's stream and do not synchronise; the others take HOST pointers,
 *     stage through device workspaces and return after the result is in `out`
 *   - re-entrant: one HIP stream + workspace pool per calling thread
 *   - float arithmetic reproduces scipy.ndimage bit for bit: double accumulation in
 *     scipy's tap order, separately rounded multiply and add (no FMA contraction),
 *     rounding to the array dtype after every axis pass
 */
#ifndef TISSUE_HIP_H
#define TISSUE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIP_API __attribute__((visibility("default")))

typedef enum {
    TIP_OK = 0,
    TIP_ERR_HIP = -1,        /* a HIP runtime call failed */
    TIP_ERR_ARG = -2,        /* bad argument (ValueError in the Python mirror) */
    TIP_ERR_NOMEM = -3,
    TIP_ERR_INDEX = -4,      /* the reference would raise IndexError (sp.py:62,68 clip upper bound) */
    TIP_ERR_UNSUPPORTED = -5,
    TIP_ERR_OVERFLOW = -6    /* caller-provided capacity too small */
} tip_status;

/* ---- lifecycle / plumbing ------------------------------------------------------------------- */
TIP_API int tip_init(int device);                 /* bind the calling thread to `device` (default 0) */
TIP_API int tip_shutdown(void);                   /* free this thread's stream + workspaces */
TIP_API int tip_last_error(char *buf, size_t n);  /* copy this thread's last error text */
TIP_API int tip_device_count(void);
TIP_API int tip_version(void);
TIP_API int tip_malloc(void **dptr, size_t bytes);
TIP_API int tip_free(void *dptr);
TIP_API int tip_memcpy_h2d(void *dst, const void *src, size_t bytes);
TIP_API int tip_memcpy_d2h(void *dst, const void *src, size_t bytes);
TIP_API int tip_memcpy_d2d(void *dst, const void *src, size_t bytes);   /* asynchronous, calling thread's stream */
/* a (height x width_bytes) block between pitched device buffers: the windows of the local-drift map (ti.py:2152-2166) */
TIP_API int tip_memcpy2d_d2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t height);
TIP_API int tip_memset(void *dst, int value, size_t bytes);
TIP_API int tip_sync(void);                       /* wait for this thread's stream */
/* Tuning and test hooks, process-wide.  The library reads the TIP_* environment variables ONCE (at first use) and   */
/* never again; afterwards a hook changes only through this call.  value NULL or "" restores the default.           */
/*   TIP_WS_TIES = exact | fast          tie policy of the watershed (default exact, see below)                      */
/*   TIP_WS_TILE, TIP_WS_OPEN = a,b, TIP_WS_CERT_FROM, TIP_WS_NO_SKIP, TIP_WS_LDS_PAD   tile flavour / schedule       */
/*   TIP_WS_DEBUG, TIP_WS_NO_ENDGAME, TIP_WS_NO_WIDE                                   counters / stall machinery    */
/*   TIP_PROJECT_EXACT_SCORE, TIP_PROJECT_GENERIC, TIP_PROJECT_UNFUSED_PREBLUR, TIP_PROJECT_UNFUSED_MASK,            */
/*   TIP_PROJECT_DEBUG, TIP_FAST_CFG = y,x, TIP_MFMA_BLOCKS_PER_CU                     projection kernel selection   */
/*   TIP_UNET_TILE8 = -1|0|1, TIP_UNET_SPB = 1|2|3, TIP_UNET_XCD_MAP = 0|1             U-Net convolution schedule    */
/*   TIP_UNET_TAIL_UNFUSED                                                            tail morphology as separate launches */
/*   TIP_MB_SMALL = pixels, TIP_MB_BATCH = generations                                 two-valued flood: one-workgroup  */
/*                                                                                    generations / host looks         */
/* None of them changes results: they select between schedules / kernels that are tested to agree bit for bit       */
/* (TIP_WS_TIES = fast is the one exception and says so in `flags`).                                                */
/* Quiescent use only: entry points read the table without a lock (aligned ints: never torn, but a call in flight   */
/* while a hook changes may run partly under each value). Set hooks while no other thread is inside the library.    */
TIP_API int tip_set_tuning(const char *name, const char *value);

/* per-kernel timing with HIP events on the library's own stream (bench.py roofline leg) */
TIP_API int tip_prof_enable(int on);
TIP_API int tip_prof_reset(void);
/* writes lines "name count total_ms\n" ; returns number of bytes needed */
TIP_API int tip_prof_report(char *buf, size_t n);

/* ---- separable Gaussian: scipy.ndimage.gaussian_filter(mode='nearest') ---------------------- */
/* replaces bim.py:389 (blur_image), called from sp.py:37,55,70,71 and ti.py:142.                */
/* `taps` are scipy's _gaussian_kernel1d values (length 2*int(4*sigma+0.5)+1, symmetric); an axis */
/* with n==0 is skipped (scipy skips sigma<=1e-15).  dtype: 0=float32, 1=float64.               */
TIP_API int tip_gaussian_taps(double sigma, double truncate, double *taps, int cap); /* libm exp; returns n */
TIP_API int tip_correlate1d_dev(const void *in, void *out, int dtype, int z, int y, int x, int axis,
                                const double *taps_host, int n);
TIP_API int tip_gaussian3d_w(const void *in, void *out, int dtype, int z, int y, int x,
                             const double *tz, int nz, const double *ty, int ny, const double *tx, int nx);
TIP_API int tip_gaussian3d_dev_w(const void *in, void *out, int dtype, int z, int y, int x,
                                 const double *tz, int nz, const double *ty, int ny, const double *tx, int nx);
TIP_API int tip_gaussian3d_f32(const float *in, float *out, int z, int y, int x,
                               double sz, double sy, double sx, double truncate);
TIP_API int tip_gaussian2d_f64(const double *in, double *out, int y, int x, double sy, double sx, double truncate);

/* OR-ed into `method` of tip_project_u16_binned[_dev]: build_manifold=True (sp.py:56-57, 87-165), the z-map grown as a
 * spiral around the score's maximum instead of the per-pixel argmax.  bin_size must be 1; min_z is not added to the
 * z-map (as upstream). */
#define TIP_PROJECT_MANIFOLD 16
/* build_continues_manifold(score) itself (sp.py:87-165): score float32 (z, y, x) -> chosen int64 (y, x); host arrays. */
TIP_API int tip_build_manifold_f32(const float *score, int z, int y, int x, int64_t *chosen);

/* ---- surface projection: sp.py:17-85 (build_manifold=False) ---------------------------------- */
/* czyx: uint16 (C,Z,Y,X).  [zlo,zhi) is the z slice sp.py:30-31 takes when max_z>0 (else 0,Z).   */
/* taps: scipy taps for sigma 0.5 (5), 1 (9), 2 (17), 30 (241); pass NULL to have them built with */
/* libm.  proj: float64 (C,Y,X) (sp.py:74 np.zeros -> float64); zmap: int64 (Y,X) = min_z+argmax. */
TIP_API int tip_project_u16(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z,
                            int ref_ch, int airyscan, int atoh_shift,
                            const double *t05, const double *t1, const double *t2, const double *t30,
                            double *proj, int64_t *zmap);
TIP_API int tip_project_u16_dev(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z,
                                int ref_ch, int airyscan, int atoh_shift,
                                const double *t05, const double *t1, const double *t2, const double *t30,
                                double *proj, int64_t *zmap);
/* diagnostics of the certified-argmax score passes on the fp16 matrix cores (csrc/tip_corr_f16.h; tests): ONE sigma-30 pass  */
/* (241 float64 taps) of a host float32 volume along y (axis 1) or x (axis 2), data bounded by `clip`; flag bit 8: a sample   */
/* beyond that range.  tip_mfma_f16_probe: out[t] = C + sum_k a[t][k] b[t][k] as one v_mfma_f32_32x32x16_f16 computes it -- */
/* where the instruction rounds is what the certified error bound counts.                                                     */
TIP_API int tip_score_pass_f16(const float *in, float *out, int z, int y, int x, int axis, const double *taps, int ntaps, float clip,
                               int *flag);
TIP_API int tip_mfma_f16_probe(const float *a, const float *b, const float *c, float *out, int ncase);
/* sp.py:39-65, bin_size > 1: the score is reduced over bin x bin blocks (skimage block_reduce with */
/* np.mean / np.var, numpy's float32 summation order) and resized back (skimage.transform.resize,   */
/* order 1) before the argmax.  method: 0 'max_averages', 1 'max_std', 2 'multi_channel' (block     */
/* variance of the reference channel x block mean of channel (ref_ch+1)%c).  bin_size 1..128.       */
TIP_API int tip_project_u16_binned(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z,
                                   int ref_ch, int method, int bin_size, int airyscan, int atoh_shift,
                                   const double *t05, const double *t1, const double *t2, const double *t30,
                                   double *proj, int64_t *zmap);
TIP_API int tip_project_u16_binned_dev(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z,
                                       int ref_ch, int method, int bin_size, int airyscan, int atoh_shift,
                                       const double *t05, const double *t1, const double *t2, const double *t30,
                                       double *proj, int64_t *zmap);
/* Spatial tiles of one frame (BASELINE config 5; tiling.py): the 95th percentile of sp.py:33-36 is a property of the */
/* WHOLE reference channel, so tiles first add up 65536-bin histograms of their interiors (tip_hist_u16_box_dev adds   */
/* the box [z0,z1) x [y0,y1) x [x0,x1) of channel ch into hist_dev, uint64 counts of the offset-corrected values), and */
/* every tile (+ halo) is then projected with the frame's histogram instead of its own.                               */
TIP_API int tip_hist_u16_box_dev(const uint16_t *czyx, int c, int z, int y, int x, int ch, int z0, int z1, int y0, int y1,
                                 int x0, int x1, int airyscan, unsigned long long *hist_dev);
TIP_API int tip_project_u16_hist_dev(const uint16_t *czyx, int c, int z, int y, int x, int zlo, int zhi, int min_z,
                                     int ref_ch, int airyscan, int atoh_shift,
                                     const double *t05, const double *t1, const double *t2, const double *t30,
                                     const unsigned long long *hist_dev, double *proj, int64_t *zmap);

/* ---- U-Net convolution epilogue (pl.py:31-37): x = relu(x + bias[c]) * scale[c] + shift[c] in place on a channels-last */
/* float32 activation of n values with c channels (c % 4 == 0), launched on `stream` (a hipStream_t taken as is: NULL is */
/* HIP's null stream, torch's default).  The convolutions themselves run in PyTorch-ROCm / MIOpen.                       */
TIP_API int tip_bias_relu_affine_f32_dev(float *x, const float *bias, const float *scale, const float *shift, long n, int c,
                                         void *stream);
/* Ordering edges between the calling thread's library stream and another HIP stream (torch's current stream); neither */
/* blocks the host.  tip_wait_stream: later library work starts after everything queued on `stream` so far -- call it  */
/* AFTER allocating every torch tensor the library is going to write (the caching allocator hands out blocks whose     */
/* previous owner's kernels may still be queued on that stream).  tip_stream_wait_tip: the other direction.            */
/* >= 0: the device ordinal THIS library's HIP runtime attributes to device pointer p; negative: unknown to it (a     */
/* second copy of libamdhip64 in the process) -- callers that pass torch pointers / streams check this once.          */
TIP_API int tip_pointer_device(const void *p);
TIP_API int tip_wait_stream(void *stream);
TIP_API int tip_stream_wait_tip(void *stream);
/* ---- the U-Net's layers (pl.py:31-72) on the 16-bit matrix cores with split float32 operands ------------------- */
/* Activations between layers are `planes` 16-bit images [plane][y][x][channel] whose sum is the float32 value:       */
/* format 0 = bf16 pieces (planes 2: three products per term, 2^-16; planes 3: six products, float32-equivalent),      */
/* format 1 = fp16 pieces (planes 2) of values SCALED by a power of two: three products per term, float32-equivalent   */
/* to 2^-21 (csrc/tip_unet_conv.h has the arithmetic and its error bound).  Every call launches on `stream` (torch's   */
/* current stream: the buffers are torch tensors) and returns at once.  Tensors may be of any size (a tile's halo      */
/* window, 18 rows of one plane, has to stay below 4 GB).                                                              */
typedef struct tip_unet_conv_desc {
    const void *in0, *in1;      /* split activations; in1 (c1 channels) is appended to in0's channels: concatenate   */
    int c0, c1, h, w, planes;   /* channels (multiples of 16), input grid (multiples of 8 x 32), pieces per value     */
    const void *weights;        /* packed split weights [tap][cin/16][cout/128][plane][128][16] bf16; in every group of 32    */
                                /* output channels row 8g + 4h + j (g < 4, h < 2, j < 4) holds channel 16h + 4g + j            */
    int ntaps, dy[9], dx[9];    /* taps: input offset (-1, 0, 1) each; Conv2D 3x3: the nine offsets in kernel order   */
    int cout;                   /* multiple of 128                                                                     */
    const float *bias, *scale, *shift;   /* scale / shift NULL: bias only (Conv2DTranspose); else bias -> ReLU -> BN */
    void *out;                  /* [plane][out_h][out_w][cout]; input-grid pixel (y, x) -> (y * sy + oy, x * sx + ox)  */
    int out_h, out_w, sy, sx, oy, ox;
    void *pool_out;             /* NULL, or [plane][out_h / 2][out_w / 2][cout]: MaxPool2D(2) of the output, written from  */
                                /* the same registers (Conv2D -> MaxPool2D, pl.py:42-43); needs sy = sx = 1, oy = ox = 0   */
    const float *head_w, *head_b; /* NULL, or the network's head fused into this layer (cout == 128, plain mapping, BN present):  */
    float *head_out;            /* Conv2D(128 -> 2, 1x1) weights [2][128], bias [2], softmax -> float32 (2, h, w); `out` unused    */
    int format;                 /* 0: bf16 pieces; 1: fp16 pieces (planes == 2) -- activations, weights and the constants carry the  */
    float acc_scale;            /* caller's power-of-two scales, and the accumulator is multiplied by acc_scale before the bias      */
    int tf;                     /* != 0: Conv2DTranspose(3, 2, 'same') as ONE launch: ntaps = 4 input offsets (0,0), (0,-1), (-1,0),      */
    int nmask[9];               /* (-1,-1); `cout`, weights and bias count VIRTUAL channels [cout / 32][class (py, px)][32]; nmask[t] = the */
                                /* classes tap t feeds (bit py * 2 + px); out = [plane][2h][2w][cout / 4], sy = sx = 2, bias only          */
} tip_unet_conv_desc;
TIP_API int tip_unet_conv_dev(const tip_unet_conv_desc *d, void *stream);
/* first layer, Conv2D(2 -> 128): float32 (2, h, w) in, weights [9][2][128] float32, exact float32 FMAs               */
TIP_API int tip_unet_conv_first_dev(const float *in, int h, int w, const float *wgt, const float *bias, const float *scale,
                                    const float *shift, void *out, int planes, int format, void *stream);
TIP_API int tip_unet_pool2_dev(const void *in, int h, int w, int ch, int planes, int format, void *out, void *stream);   /* MaxPool2D(2) */
/* Conv2D(128 -> 2, 1x1) + softmax: float32 (2, npix) out; logits != 0: the pre-softmax values                         */
TIP_API int tip_unet_head_dev(const void *in, long npix, const float *wgt, const float *bias, float *out, int planes, int format,
                              int logits, void *stream);
/* U1, prepare_image + normalize_channel (pl.py:21-29, 90-122) on a device-resident image: img = (c, a, b) float64,       */
/* element strides (cstride, sa, sb), every channel plane dense in either orientation; kind = dtype of the caller's image  */
/* (0 float64, 1 float32, 2 integer: the clip values take it, pl.py:26-27).  Per channel: np.percentile 1 / 99 (exact order  */
/* statistics by radix select + numpy's lerp), clip, scale; out = (c, bp, ap) float32, transposed, zero-padded in front.      */
TIP_API int tip_unet_prepare_f64_dev(const double *img, int c, int a, int b, long cstride, long sa, long sb, int kind, float *out,
                                     int ap, int bp, void *stream);
/* pl.py:167-194 after the network, one submission: p = class-0 probability map on the device (y rows of x values, row  */
/* pitch ld elements; dtype 0 = float32, 1 = float64) -> 255 (p > thr) -> 5x5 closing -> HC = 7x7 erosion -> boundary = */
/* 5x5 dilation of (closed - HC) -> watershed(watershed_line=True).  labels / hc: caller-owned device buffers (y * x).  */
/* A boundary image that is not two-valued is an error (corrupted intermediate), never a slow flood.                   */
TIP_API int tip_unet_tail_dev(const void *p, int dtype, long ld, int y, int x, double thr, int32_t *labels, double *hc,
                              int32_t *flags_host);

/* ---- overlay images of Tissue.draw_* (ti.py:584-607, 2585-2645): (3, y, x) float64 images the GUI composites over a frame.   */
/* Host arrays in and out.  cell types: positive = all bits of must_mask set (and the byte != 255 unless must_mask is 0) and   */
/* no bit of lack_mask set on a valid byte (is_positive_for_type, ti.py:146-176); negative = valid and not positive.          */
/* tracking: colour cycle18[id % 6], id 0 black (ti.py:2625-2635).  disks: skimage.draw.disk(center, radius, shape), a later    */
/* disc paints over an earlier one (draw_events / draw_cell_tracking / draw_marking_points).  lines: skimage.draw.line          */
/* between (r0, c0, r1, c1) quadruples (draw_neighbors_connections).                                                            */
TIP_API int tip_draw_cell_types_u8(const uint8_t *types, long n, int must_mask, int lack_mask, const double *pos_rgb,
                                   const double *neg_rgb, double *out3);
TIP_API int tip_draw_tracking_i32(const int32_t *track, long n, const double *cycle18, double *out3);
TIP_API int tip_draw_disks_f64(int y, int x, int n, const double *cy, const double *cx, double radius, const double *rgb, double *out3);
TIP_API int tip_draw_lines_f64(int y, int x, int n, const int32_t *ends, const double *rgb, double *out3);

/* ---- rank filters ---------------------------------------------------------------------------- */
/* scipy.ndimage.maximum_filter / minimum_filter (ti.py:1822,2081,2969,4079-4084) and             */
/* skimage.morphology.erosion/dilation with a flat footprint (pl.py:170-193).                     */
/* footprint_kind: 0 = full ky x kx rectangle, 1 = 3x3 cross without centre ([[0,1,0],[1,0,1],[0,1,0]]). */
/* border_mode: 0 = constant 0, 1 = reflect.  is_max: 1 max / 0 min.                              */
TIP_API int tip_rankfilter2d(const void *in, void *out, int dtype /*1=f64, 2=i32*/, int y, int x, int ky, int kx,
                             int footprint_kind, int border_mode, int is_max);
TIP_API int tip_rankfilter2d_dev(const void *in, void *out, int dtype, int y, int x, int ky, int kx,
                                 int footprint_kind, int border_mode, int is_max);
/* bim.py:464-473: thr = imgthresh*max_filter(img, block, reflect) ; out = img < thr ? 0 : img (float64) */
TIP_API int tip_local_threshold_f64_dev(const double *img, double *out, int y, int x, double imgthresh, int block);

/* ---- connected components: skimage.measure.label(connectivity=1) (ti.py:2922,3470) ----------- */
/* equal-valued 4-neighbours are connected, `bg` pixels -> 0, labels 1..n in raster order of the  */
/* component's first pixel.  Also scipy.ndimage.label on a boolean image (watershed markers).     */
TIP_API int tip_label4_i32(const int32_t *in, int32_t bg, int32_t *out, int y, int x, int32_t *n_labels);
TIP_API int tip_label4_i32_dev(const int32_t *in, int32_t bg, int32_t *out, int y, int x, int32_t *n_labels_host);

/* ---- watershed: skimage.segmentation.watershed(markers=None, connectivity=1) (bim.py:475, pl.py:194) */
/* markers = label(local_minima(img)).  The result equals skimage's bit for bit.  Three routes, reported in `flags`  */
/* (out, may be NULL):                                                                                               */
/*   - landscapes whose non-marker pixels carry distinct values: the data-parallel certified flood (no flag);        */
/*   - two-valued images (pl.py:194): TIP_WS_FLAG_TWO_VALUED, the generation-ranked flood, exact;                    */
/*   - other landscapes with value ties (TIP_WS_FLAG_TIES; the uint16 frames of gui.py:1841-1845): skimage's result */
/*     is a function of its heap array's history, so the flood itself runs as the exact serial (value, age) replay  */
/*     on one host core (TIP_WS_FLAG_SERIAL_EXACT; ~0.3 us per pixel), markers before and everything after on the   */
/*     device.  tip_set_tuning("TIP_WS_TIES", "fast") keeps such images on the device instead: ties are then broken */
/*     by raster index, not by push age, and the labels differ from skimage's in a fraction of the pixels.          */
/* TIP_WS_FLAG_SERIAL_FINISH: the device flood stalled on a serial dependency chain (plateaus larger than any       */
/* certificate; only with the fast tie policy) and the rest was finished on the host; the number of pixels decided  */
/* there is flags >> TIP_WS_FLAG_COUNT_SHIFT (saturating at 2^23 - 1).                                              */
#define TIP_WS_FLAG_TIES 1
#define TIP_WS_FLAG_TWO_VALUED 2
#define TIP_WS_FLAG_SERIAL_EXACT 4
#define TIP_WS_FLAG_SERIAL_FINISH 8
#define TIP_WS_FLAG_COUNT_SHIFT 8
TIP_API int tip_watershed_f64(const double *img, int32_t *labels, int y, int x, int wsl, int32_t *flags);
TIP_API int tip_watershed_f64_dev(const double *img, int32_t *labels, int y, int x, int wsl, int32_t *flags_host);
/* Pop order of m equal-keyed heap entries pushed in raster order when popping entry i is followed  */
/* by c[i] pushes of larger entries (skimage's heap_general.pxi mechanics; the marker phase of      */
/* pl.py:194): e[i] = position of entry i in the pop order.  Host arrays, no device involved.      */
TIP_API int tip_marker_pop_order_host(const uint8_t *c, long m, uint32_t *e);
/* The serial (value, age) flood by itself: host arrays, markers given (> 0 = seed), no device involved.            */
TIP_API int tip_watershed_serial_host(const double *img, const int32_t *markers, int32_t *labels, int y, int x);
/* number of labels (= markers) produced by the calling thread's last watershed call                 */
TIP_API int tip_last_watershed_labels(void);
/* bim.py:446-476 as one device pipeline: local threshold -> Gaussian(sigma) -> watershed          */
TIP_API int tip_watershed_segmentation_f64_dev(const double *img, int32_t *labels, int y, int x, double imgthresh,
                                               const double *taps, int ntaps, int block, int32_t *flags_host);

/* ---- cell tables: regionprops_table + find_neighbors (ti.py:880-909,1815-1842) --------------- */
/* SoA outputs over labels 1..n: area, bbox(min_row,min_col,max_row+1,max_col+1), coordinate sums  */
/* (centroid = sum/area), perimeter code counts pc[3*l+{0,1,2}] (weights 1, sqrt2, (1+sqrt2)/2),   */
/* optional intensity sums.                                                                        */
TIP_API int tip_regionprops_i32(const int32_t *labels, const double *intensity /*nullable*/, int y, int x, int n,
                                int64_t *area, int64_t *bbox4, int64_t *sumy, int64_t *sumx, int64_t *pc3,
                                double *isum /*nullable*/);
TIP_API int tip_regionprops_i32_dev(const int32_t *labels, const double *intensity, int y, int x, int n,
                                    int64_t *area, int64_t *bbox4, int64_t *sumy, int64_t *sumx, int64_t *pc3,
                                    double *isum);
/* unique (hi,lo) pairs: a pixel labelled lo>0 whose zero-padded 5x5 maximum is hi != lo           */
TIP_API int tip_neighbor_pairs_i32(const int32_t *labels, int y, int x, int32_t *pairs, int64_t cap, int64_t *n_pairs);
TIP_API int tip_neighbor_pairs_i32_dev(const int32_t *labels, int y, int x, int32_t *pairs_dev, int64_t cap,
                                       int64_t *n_pairs_host);
/* Exact order statistics per label (calc_cell_types' per-cell np.percentile, ti.py:2349-2355) by radix select:       */
/* lo[l] = the value of 0-based rank ranks[l] among the pixels of label l+1 of img (float64), hi[l] = the value of rank */
/* ranks[l]+1 (= lo[l] when the label has no such pixel); ranks[l] < 0 skips label l+1.  labels == NULL: nlab must be 1 */
/* and the statistic is taken over the whole frame (np.percentile(img, 99), ti.py:2371).                               */
TIP_API int tip_label_order_stats_f64(const int32_t *labels, const double *img, int y, int x, int nlab, const int64_t *ranks,
                                      double *lo, double *hi);
/* Contact lengths (ti.py:1844-1872, 4073-4094): for every ordered label pair (hi > lo >= 1) the number of pixels whose  */
/* 4-neighbour maximum of the labels is hi and whose 4-neighbour minimum of the labels with zeros replaced by `big`    */
/* (= max label + 1, ti.py:4081) is lo; filters as scipy's with the cross footprint and mode='constant'.  pairs: (hi, */
/* lo) rows, counts: the pixel numbers, in no particular order.                                                        */
TIP_API int tip_contact_pairs_i32(const int32_t *labels, int y, int x, int big, int32_t *pairs, int64_t *counts, int64_t cap,
                                  int64_t *n_pairs);
/* Tissue.update_labels (ti.py:2967-2970): negatives take the zero-padded 3x3 maximum              */
TIP_API int tip_update_labels_i32(int32_t *labels, int y, int x);
/* track_cells_iterator's label lookup (ti.py:2081-2090): maximum_filter(labels,(3,3),'constant') sampled at query   */
/* points (host arrays); out[i] = -1 for points outside the frame. labels is a DEVICE pointer.                     */
TIP_API int tip_lookup_max3_i32_dev(const int32_t *labels, int y, int x, const int64_t *qy_host, const int64_t *qx_host,
                                    int64_t n, int32_t *out_host);
/* Tissue.get_trackking_labels (ti.py:4021-4028): out[p] = lut[labels[p]] (lut[0] = 0)              */
TIP_API int tip_lut_gather_i32(const int32_t *labels, const int64_t *lut, int64_t n_lut, int64_t *out, int64_t n);

/* ---- drift: skimage.registration.phase_cross_correlation(ref, mov, upsample_factor) ---------------------------- */
/* (ti.py:1976-1977, 2029-2030 update_drift / calculate_refine_drift; bim.py:522-536 calculate_drift).               */
/* dtype: 0 float32, 1 float64, 3 uint16; extents in [2, 4096] (powers of two: radix-2 FFT rows; anything else: Bluestein).  out4 = whole-pixel peak (row, col) of       */
/* |ifft2(F1 conj F2)| and the peak (row, col) on the ceil(1.5*upsample)^2 upsampled grid; the caller forms the shift  */
/* exactly as skimage does (wrap past the midpoint, round to the grid, add (fine - floor(region/2)) / upsample).       */
TIP_API int tip_phase_correlation(const void *ref, const void *mov, int dtype, int y, int x, int upsample, int64_t *out4);
TIP_API int tip_phase_correlation_dev(const void *ref, const void *mov, int dtype, int y, int x, int upsample,
                                      int64_t *out4_host);

#ifdef __cplusplus
}
#endif
#endif
