/*
 * ccnet_cca.h -- C ABI of the MI355X-native criss-cross attention library (libccnet_cca.so).
 *
 * This is the drop-in boundary for CCNet's hot path.  The mounted reference is the pure-python
 * branch, so there is no FFI in the tree to bind; each entry point below replaces a chain of
 * torch ops in /root/reference/cc_attention/functions.py (cited per function) and carries the
 * name the reference's CUDA-extension branches use for the same quantity (ca_forward /
 * ca_backward / ca_map_forward / ca_map_backward, BASELINE.json north_star).  The reference-side
 * binding a maintainer would add (a ctypes stub in cc_attention/functions.py) is in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator in the Python host).  The library never allocates or frees.
 *   - feature maps are NCHW-contiguous fp32:  q,k (B,Cq,H,W);  v,x,y and their grads (B,C,H,W).
 *   - attention-shaped tensors (energy, A, dA, dE) are (B,H,W,H+W) contiguous, slot-fastest, in
 *     the reference's ``concate`` order (functions.py:40): slots [0,H) are the column branch
 *     (key/value at (j,w); slot j==h is the masked self slot), slots [H,H+W) the row branch
 *     (key/value at (h,j)).
 *   - outputs are fully overwritten; nothing relies on pre-zeroed buffers.
 *   - launches go to ``stream`` (a hipStream_t; NULL = the legacy default stream); no call
 *     synchronises (the pixel-major / split-plane backwards fork part of their launches onto a library-owned side stream
 *     and join it again before they return: see "planes_overlap").  The compute entry points hold no per-call state and may be called concurrently
 *     from several host threads / streams; the only process-wide state are the three MODE words behind
 *     the options "impl", "precision", "branch_mask" of ccnet_cca_set_option (atomics; defaults need no call).  A setter
 *     racing with a call in flight affects that call or the next one, never part of one.  The fused
 *     ccnet_cca_forward / backward entry points refuse to run under a profiling branch mask.
 *   - return value: 0 on success; a positive hipError_t if a launch failed; a negative
 *     CCNET_E_* code for argument errors.  ccnet_cca_last_error_string() describes the last
 *     failure on the calling thread.
 *   - ``gamma`` is always a DEVICE pointer to one float (the module's nn.Parameter), so no
 *     host synchronisation is needed to read it.
 */
#ifndef CCNET_CCA_H
#define CCNET_CCA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the functions declared in this header -- and nothing else -- are its dynamic
 * symbols (tests/test_host.py compares ``nm -D`` with this header, both ways). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define CCNET_CCA_VERSION 220          /* 0.2.2.  Over 0.2.0: ccnet_cca_pack_projection_f32, ccnet_cca_probe_*, ccnet_cca_backward_planes3_f32, ccnet_cca_projection_bf16 / _adjoint_bf16 / _wgrad_bf16; only ccnet_* symbols are exported.
                                          BEHAVIOURAL changes a binding must know (0.2.1 called these "additive"):
                                          - the *_BACKWARD workspaces of the pixel-major / split-plane entry points are 256 B larger than in 0.2.0
                                            (a binding that hard-coded the 0.2.0 formula gets CCNET_E_WORKSPACE: query the size, as always);
                                          - fp32 ca_backward (dq | dk) of those entry points multiplies as SIX bf16 terms (fp32-equivalent) instead of
                                            three -- option "dqdk_exact" is now 0 / 1, default 1; the values 1 = exact-f32 MFMA and 2 = device-gated
                                            redo of 0.2.1 are gone (CCNET_E_BADFLAGS);
                                          - the bf16 entry points round the column -> row partial to bf16 (option "bf16_partial", default 1). */

#define CCNET_E_BADSHAPE   (-1)        /* non-positive dimension, or a size the kernels cannot index */
#define CCNET_E_NULLPTR    (-2)        /* a required pointer is NULL */
#define CCNET_E_BADFLAGS   (-3)
#define CCNET_E_WORKSPACE  (-4)        /* workspace missing or too small */

/* ccnet_ca_forward flags */
#define CCNET_CA_ENERGY    0           /* write raw energies, masked slot = -inf (functions.py:38-40, pre-softmax) */
#define CCNET_CA_SOFTMAX   1           /* write A = softmax over the H+W slots (functions.py:40) */

/* kernel family selection (debug / A-B testing; default AUTO picks MFMA when the shape fits) */
#define CCNET_IMPL_AUTO    0
#define CCNET_IMPL_DIRECT  1           /* one-thread-per-output kernels, any shape */
#define CCNET_IMPL_MFMA    2           /* MFMA strip kernels only (max(H,W) <= 320), error beyond */

/* Arithmetic of the NCHW STRIP kernels (ccnet_ca_*_f32, ccnet_cca_*_f32, *_strided_f32, *_ws_f32).  The affinity
 * (ca_forward), the softmax and the dq/dk kernels of that family always run exact fp32 (the f32 MFMA is bit-identical
 * to an fmaf chain).  The PIXEL-MAJOR and SPLIT-PLANE entry points (*_pm_*, *_planes_*) are not governed by this knob:
 * they compute the energies in exact fp32, ca_backward (dq | dk) as six bf16 terms (fp32-equivalent) and every other contraction as
 * split-bf16 x3 (NUMERICS ENVELOPE below, at ccnet_cca_forward_planes_f32) -- callers that
 * pin CCNET_PRECISION_F32 or CCNET_IMPL_DIRECT for validation must call the strip / direct entry points; the Python module
 * reads the two knobs (ccnet_cca_get_option) and routes accordingly.
 * The two aggregation-type contractions of the strip family (ca_map_forward, ca_map_backward's dv) may split every fp32 operand
 * into bf16 hi + lo and evaluate the products on the bf16 matrix pipe with fp32 accumulation (relative error ~2^-17 per product;
 * measured max-abs error at (8,512,97,97): 3e-5 on y / dv, inside the 1e-3 fp32 parity bar); its affinity, dA and dq / dk kernels
 * are exact fp32 in every mode (round 4 retired the packed split-bf16 dA kernel: it spilled 126 VGPRs):
 *   F32      exact fp32 everywhere
 *   DEFAULT  split-bf16 in the ROW launches of the aggregation kernels; exact fp32 in their column launches (no gain there)
 *   BF16X3   split-bf16 in both launches of the aggregation kernels
 * The split variants exist for strips 97..100 long; other shapes run exact fp32. */
#define CCNET_PRECISION_F32     0
#define CCNET_PRECISION_BF16X3  1
#define CCNET_PRECISION_DEFAULT 2

/* profiling aid: restrict the strip-kernel launches of every entry point to one branch so a single
 * kernel can be timed in isolation (results are then partial).  Default CCNET_BRANCH_BOTH. */
#define CCNET_BRANCH_COL   1
#define CCNET_BRANCH_ROW   2
#define CCNET_BRANCH_BOTH  3

typedef void *ccnet_stream_t;          /* hipStream_t */

int         ccnet_cca_version(void);
const char *ccnet_cca_arch(void);                  /* "gfx950" */
const char *ccnet_cca_last_error_string(void);
/* The MODE words above are options "impl", "precision" and "branch_mask" of ccnet_cca_set_option / ccnet_cca_get_option
 * (declared with the other options below): an invalid value is rejected (CCNET_E_BADFLAGS) and leaves the word unchanged. */

/* Scratch sizes of the entry points that take a ``workspace`` (bytes; 0 = none needed).  ``entry``: */
#define CCNET_WS_SOFTMAX_BACKWARD 0    /* ccnet_ca_softmax_backward_f32 with dgamma (C, Cq ignored) */
#define CCNET_WS_FORWARD          1    /* ccnet_cca_forward_ws_f32 */
#define CCNET_WS_BACKWARD         2    /* ccnet_cca_backward_f32 / _backward_strided_f32 */
#define CCNET_WS_PM_FORWARD       3    /* ccnet_cca_forward_pm_{bf16,f32} */
#define CCNET_WS_PM_BACKWARD      4    /* ccnet_cca_backward_pm_{bf16,f32} */
#define CCNET_WS_PLANES_FORWARD   5    /* ccnet_cca_forward_planes_f32 */
#define CCNET_WS_PLANES_BACKWARD  6    /* ccnet_cca_backward_planes_f32 */
#define CCNET_WS_SPLIT_COLSUM     7    /* ccnet_cca_split_planes_colsum_f32 (Cq ignored) */
#define CCNET_WS_PLANES3_BACKWARD 8    /* ccnet_cca_backward_planes3_f32 */
size_t      ccnet_cca_workspace_bytes(int entry, int B, int C, int Cq, int H, int W);

/* Affinity: replaces functions.py:30-34 (layout shuffles), :38 (bmm + INF), :39 (bmm), :40 (cat
 * [+ Softmax when CCNET_CA_SOFTMAX]).  out (B,H,W,H+W). */
int ccnet_ca_forward_f32(const float *q, const float *k, float *out,
                         int B, int Cq, int H, int W, int flags, ccnet_stream_t stream);

/* Adjoint of the affinity (autograd of functions.py:38-39): dE (B,H,W,H+W) -> dq, dk (B,Cq,H,W).
 * dE at the masked column self slot must be 0 (it is for every softmax-derived gradient, A == 0 there). */
int ccnet_ca_backward_f32(const float *dE, const float *q, const float *k, float *dq, float *dk,
                          int B, int Cq, int H, int W, ccnet_stream_t stream);

/* Softmax over the H+W slots in isolation (functions.py:40), in place allowed (out == energy). */
int ccnet_ca_softmax_forward_f32(const float *energy, float *out,
                                 int B, int H, int W, ccnet_stream_t stream);

/* Adjoint of the softmax: dE = g * A * (dA - sum_s A dA) with g = *gamma (1 if gamma == NULL);
 * if dgamma != NULL also writes dgamma[0] = sum A*dA (= sum dy*(out_H+out_W), the gradient of
 * functions.py:49's gamma when dA is the un-scaled map adjoint).  dE may alias dA.
 * workspace: ccnet_cca_workspace_bytes(CCNET_WS_SOFTMAX_BACKWARD, ...) bytes when dgamma != NULL. */
int ccnet_ca_softmax_backward_f32(const float *A, const float *dA, const float *gamma, float *dE,
                                  float *dgamma, void *workspace, size_t workspace_bytes,
                                  int B, int H, int W, ccnet_stream_t stream);

/* Aggregation: replaces functions.py:36-37,42,45 (layout shuffles), :46-47 (bmm x2), :49 (epilogue).
 * out = g * (out_H + out_W) + x, with g = *gamma (1 if NULL) and x optional (NULL -> no residual).
 * With gamma == NULL and x == NULL this is the plain ca_map_forward of the extension API.
 * out must not alias x or v (the two launches of the pair pass partial sums through out). */
int ccnet_ca_map_forward_f32(const float *A, const float *v, const float *x, const float *gamma,
                             float *out, int B, int C, int H, int W, ccnet_stream_t stream);

/* Adjoint of the aggregation (autograd of functions.py:42-47):
 *   dA (B,H,W,H+W) = un-scaled map adjoint  (sum_c dout * v at the slot's source pixel); may be NULL
 *   dv (B,C,H,W)   = g * (A^T-weighted sums of dout), g = *gamma (1 if NULL); may be NULL */
int ccnet_ca_map_backward_f32(const float *dout, const float *A, const float *v, const float *gamma,
                              float *dA, float *dv, int B, int C, int H, int W, ccnet_stream_t stream);

/* Workspace sizes of the fused entry points.  The workspace is optional scratch that lets SMALL BATCHES (the
 * reference trains at 1-2 images per GPU: README.md:97, engine.py:88) use the whole chip: the channel contractions
 * behind the affinity (forward) and behind dA (backward) are split into channel ranges whose partial attention-shaped
 * results live in the workspace and are added, in a fixed order, by the softmax kernels.  With a NULL / smaller
 * workspace the same results are computed unsplit (backward: at least
 * CCNET_WS_SOFTMAX_BACKWARD is always required; the backward size includes it).  At batch sizes
 * that fill the chip the forward size is 0 and the backward size is just the softmax part. */

/* The strided forward (see "strided forms" above), with the optional workspace described above (NULL / 0: unsplit). */
int ccnet_cca_forward_ws_f32(const float *q, const float *k, const float *v, const float *x, const float *gamma,
                             float *y, float *A, int B, int C, int Cq, int H, int W,
                             long q_bs, long k_bs, long v_bs, void *workspace, size_t workspace_bytes,
                             ccnet_stream_t stream);

/* The attention tensor alone: A = softmax(ca_forward(q, k)) with q, k addressed through batch strides (channel
 * slices of a stacked projection).  This is what ccnet_cca_forward_ws_f32 leaves in ``A``; the host calls it in
 * the backward pass when it chose NOT to keep A between forward and backward (recompute instead of save:
 * SURVEY.md 8(f) rank 4, networks/ccnet.py:118-119 -- R applications of the module hold R attention tensors). */
int ccnet_cca_attention_strided_f32(const float *q, const float *k, float *A, int B, int Cq, int H, int W,
                                    long q_bs, long k_bs, ccnet_stream_t stream);

/* Fused core, forward: (q,k,v,x,gamma) -> y and the saved attention A  (functions.py:38-49). */
int ccnet_cca_forward_f32(const float *q, const float *k, const float *v, const float *x,
                          const float *gamma, float *y, float *A,
                          int B, int C, int Cq, int H, int W, ccnet_stream_t stream);

/* Fused core, backward: dy + saved (q,k,v,A,gamma) -> dq, dk, dv, dgamma (dx == dy is the caller's).
 * ``scratch`` is an attention-shaped (B,H,W,H+W) fp32 buffer; ``workspace`` as for softmax_backward. */
int ccnet_cca_backward_f32(const float *dy, const float *q, const float *k, const float *v,
                           const float *A, const float *gamma, float *dq, float *dk, float *dv,
                           float *dgamma, float *scratch, void *workspace, size_t workspace_bytes,
                           int B, int C, int Cq, int H, int W, ccnet_stream_t stream);

/* Strided forms of the fused core: q, k, v (and dq, dk, dv) may be channel slices of ONE (B, 2*Cq+C, H, W)
 * projection -- the output of a single fused 1x1 convolution that replaces functions.py:29,32,35 -- so no
 * copy is needed to make them contiguous.  Each still is a dense (C, H, W) image per batch element; the
 * ``*_bs`` arguments are the distance between consecutive batch elements in ELEMENTS (dense tensor: C*H*W;
 * a slice of the fused projection: (2*Cq+C)*H*W).  Base pointers must be 4-byte aligned.  x, y, dy, A and
 * scratch stay dense.  A stride below C*H*W returns CCNET_E_BADSHAPE. */
int ccnet_cca_backward_strided_f32(const float *dy, const float *q, const float *k, const float *v,
                                   const float *A, const float *gamma, float *dq, float *dk, float *dv,
                                   float *dgamma, float *scratch, void *workspace, size_t workspace_bytes,
                                   int B, int C, int Cq, int H, int W,
                                   long q_bs, long k_bs, long v_bs, long dq_bs, long dk_bs, long dv_bs,
                                   ccnet_stream_t stream);

/* Pixel-major bf16 path (BASELINE.json configs[4], (16,512,129,129) bf16): every feature tensor is a bf16
 * (B, H*W, pixel stride) VIEW -- element (b, c, h, w) at  b * bs + (h * W + w) * ps + c  (units: elements) -- which is
 * what a channels_last tensor is, and what the reference module's 1x1 projections (functions.py:29,32,35) produce when they run
 * as one  x^T W^T  GEMM: q, k, v are then channel slices of ONE (B, H*W, 2*Cq + C) projection and no copy is made.
 * The attention tensor A (B, H, W, H+W), the scratch tensor, gamma, dgamma and every accumulation are fp32; products
 * of the bf16 features are exact on the matrix pipe; outputs are rounded to nearest even once, on store.
 * Constraints: max(H, W) <= 132, C % 8 == 0, Cq % 8 == 0, every bs / ps a multiple of 8, pointers 16-byte aligned.
 * y = gamma * (column + row aggregation) + x       (functions.py:46-49)
 * ``workspace``: ccnet_cca_workspace_bytes(CCNET_WS_PM_FORWARD / _BACKWARD, ...) bytes (fp32 column partials; + softmax
 * partials; bf16 and fp32 views alike). */
int ccnet_cca_forward_pm_bf16(const uint16_t *q, const uint16_t *k, const uint16_t *v, const uint16_t *x,
                              const float *gamma, uint16_t *y, float *A, int B, int C, int Cq, int H, int W,
                              long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long x_bs, int x_ps,
                              long y_bs, int y_ps, void *workspace, size_t workspace_bytes, ccnet_stream_t stream);
/* ``scratch``: B*H*W*(H+W) floats (dA, then dE in place). */
int ccnet_cca_backward_pm_bf16(const uint16_t *dy, const uint16_t *q, const uint16_t *k, const uint16_t *v,
                               const float *A, const float *gamma, uint16_t *dq, uint16_t *dk, uint16_t *dv,
                               float *dgamma, float *scratch, int B, int C, int Cq, int H, int W,
                               long dy_bs, int dy_ps, long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps,
                               long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                               void *workspace, size_t workspace_bytes, ccnet_stream_t stream);

/* The same core on fp32 pixel-major views (strips <= 100, C % 4 == 0, Cq % 4 == 0, every bs / ps a multiple of 4): one
 * strip per workgroup instead of 8 per workgroup -- 26x more workgroups per launch, which is what 1-2 images per GPU
 * need (HISTORY.md 3.8).  fp32 features are split into bf16 hi + lo on the fly (the three-product form of HISTORY.md 3.7);
 * same arguments, semantics and workspace as the bf16 pair.  (What the module runs for channels_last fp32 inputs; NCHW fp32
 * inputs take the split-plane path below.) */
int ccnet_cca_forward_pm_f32(const float *q, const float *k, const float *v, const float *x,
                             const float *gamma, float *y, float *A, int B, int C, int Cq, int H, int W,
                             long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long x_bs, int x_ps,
                             long y_bs, int y_ps, void *workspace, size_t workspace_bytes, ccnet_stream_t stream);
int ccnet_cca_backward_pm_f32(const float *dy, const float *q, const float *k, const float *v,
                              const float *A, const float *gamma, float *dq, float *dk, float *dv,
                              float *dgamma, float *scratch, int B, int C, int Cq, int H, int W,
                              long dy_bs, int dy_ps, long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps,
                              long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                              void *workspace, size_t workspace_bytes, ccnet_stream_t stream);

/* Which kernel family serves this shape under the current impl setting: 1 = stationary MFMA strip kernels
 * (max(H,W) <= 100), 2 = windowed MFMA strip kernels (101 .. 320), 0 = any-shape kernels. */
int ccnet_cca_shape_uses_mfma(int B, int C, int H, int W);

/* ---- SPLIT-PLANE path: the fp32 core with its C-sized contraction operands stored PRE-SPLIT ----
 * A "planes" tensor is an fp32 feature tensor stored as two bf16 planes per pixel, (B, H*W, 2, C) uint16:
 *     hi = bf16_rne(x) at [b][p][0][c],  lo = bf16_rne(x - hi) at [b][p][1][c]   (x = hi + lo + O(2^-17 |x|))
 * -- the same bytes as fp32, pixel-major, split ONCE by the producer, so that the kernels that contract over it
 * (functions.py:42-47 and their adjoints) run the bf16 matrix pipe with three exact products per term and no per-use split.
 * Views: pointer to the hi plane, batch stride and pixel stride in ELEMENTS (pixel stride >= 2 C, both multiples of 8);
 * the lo plane of a pixel starts C elements after its hi plane.  C % 8 == 0, max(H, W) <= 132 (strips up to 100 run the
 * kernels tuned for the headline geometry; 101 .. 132 -- the 129 x 129 map of BASELINE configs[4] in fp32 -- the same kernels
 * padded to 132 positions, the row passes with one workgroup per CU).
 *
 * ccnet_cca_split_planes_f32: fp32 pixel-major view (e.g. the value slice of the packed projection x^T W^T) -> planes;
 *   ``bias`` (C floats, or NULL) is added to every pixel before the split (the projection's bias, functions.py:35).
 * ccnet_cca_nchw_to_planes_f32: NCHW fp32 (B, C, H, W) -> planes.
 *   ``layout`` of both: CCNET_PLANES_HL = hi | lo as above (what the core consumes).  The THREE-plane rows are operands of
 *   K-concatenated split-bf16 GEMMs on a stock bf16 -> fp32 GEMM (the projections either side of the core, functions.py:29-35:
 *   x.w ~ xh.wh + xh.wl + xl.wh is ONE GEMM with K = 3C on rows [xh | xh | xl] x [wh | wl | wh]): CCNET_PLANES_HLH =
 *   hi | lo | hi, CCNET_PLANES_HHL = hi | hi | lo (pixel stride >= 3 C).  Paired row by row, HLH x HHL gives the three
 *   products as well (the weight gradient as one GEMM over 3 B H W rows).
 * LONG STRIPS: both entry points also take maps whose rows, columns or both have 133 .. 528 positions (max(H, W) <= 528,
 *   C/8 <= 64) -- the 129 x 257 map of the reference's whole-image evaluation (evaluate.py:102-143) and the 193 x 385 ..
 *   257 x 513 maps of its multi-scale variant (evaluate.py:146-166, scales 1.5 .. 2).  A long strip is cut into blocks of
 *   <= 132 positions: the energies / dA kernels compute one (query block, key block) tile pair per workgroup; every pass of
 *   a branch with long strips (aggregation, dv, dq | dk) runs once per block of the CONTRACTED positions, updating its fp32
 *   partial in place -- a workgroup owns one OUTPUT block of a strip -- and the last row launch writes the output.
 * ccnet_cca_forward_planes_f32 / _backward_planes_f32: functions.py:38-49 and its autograd with q, k fp32 pixel-major
 *   views (exact fp32 energies), the module's x / y / dy NCHW fp32, dq | dk | dv fp32 pixel-major views, A / scratch
 *   (B,H,W,H+W) fp32 as everywhere.  The forward takes v the way its producer leaves it -- ``v``: the fp32 pixel-major value
 *   slice of the projection (functions.py:35), ``v_bias``: C floats added while splitting, or NULL -- and WRITES ``v_planes``
 *   (its first launch), which the caller keeps for the backward; with ``v`` == NULL, ``v_planes`` is an input that already
 *   holds the planes.  PLANE-FREE FORM (``v_planes`` == NULL, strips <= 100, no ``v_bias``): v is consumed as the fp32 tensor it
 *   is -- the aggregation and (backward: pass the same ``v``, ``v_planes`` == NULL) the dA contraction split every fragment into
 *   bf16 hi | lo in registers; no planes tensor, no split pass (308 MB less traffic per forward); same arithmetic, same bits.
 *   Workspace: CCNET_WS_PLANES_FORWARD / _BACKWARD (backward: holds the fp32 column partial and dy as planes).
 *   Arithmetic: energies exact fp32; ca_backward (dq | dk from dE) as SIX bf16 terms of a three-way split (fp32-equivalent, 2^-24);
 *   every other contraction -- aggregation, dv, dA -- split-bf16 x3 with fp32 accumulation (the lo x lo term, 2^-18 relative, is
 *   dropped; dy is consumed as two bf16 planes, 2^-17 relative per element) -- the CCNET_PRECISION_* knob does not apply here.
 *   NUMERICS ENVELOPE (asserted by tests/test_gpu_parity.py: logit-, value- and gradient-scale sweeps at (.,512,97,97), hot-logit maps
 *   of 129 x 129 and 129 x 257): every output O in {y, dv, dq, dk} satisfies  max |O - O_ref| <= 2e-5 * max |O_ref|  against the fp64-
 *   accumulating restatement of functions.py:27-49 (measured: 4e-6 .. 1e-5), at any scale of q, k, v, dy.  The ABSOLUTE 1e-3 bar of
 *   the north_star therefore holds wherever max |O_ref| <= 50 -- N(0,1) inputs at the reference's geometry give max |y|, |dv| ~ 6 and
 *   max |dq|, |dk| ~ 49; q, k three times hotter (|dq| ~ 136) still measure 6.4e-4 -- and not for arbitrarily large |v| * |dy|
 *   (v, dy x 4 each: |dq| ~ 780, error 4e-3 = 5.5e-6 relative: the 2^-17 of the three-term dA).  The exact-fp32 family
 *   (ccnet_ca_*_f32 on NCHW q, k, v) is there for callers that need fp32 products throughout. */
#define CCNET_PLANES_HL 2
#define CCNET_PLANES_HLH 3
#define CCNET_PLANES_HHL 4
int ccnet_cca_split_planes_f32(const float *src, uint16_t *dst, int B, int C, int H, int W, long src_bs, int src_ps,
                               long dst_bs, int dst_ps, int layout, const float *bias, ccnet_stream_t stream);
/* ccnet_cca_split_planes_f32 (no bias) that ALSO returns the column sums of its source in the same pass: ``colsum`` (C floats) =
 * sum over all images and pixels of src[.., c], added in a fixed order (deterministic).  The module's backward needs dqkv as
 * three-plane rows for its two GEMMs and its sum over pixels as the bias gradients (functions.py:29,32,35): one pass over dqkv
 * instead of two.  ``workspace``: ccnet_cca_workspace_bytes(CCNET_WS_SPLIT_COLSUM, B, C, 0, H, W) bytes. */
int ccnet_cca_split_planes_colsum_f32(const float *src, uint16_t *dst, float *colsum, void *workspace, size_t workspace_bytes,
                                      int B, int C, int H, int W, long src_bs, int src_ps, long dst_bs, int dst_ps, int layout,
                                      ccnet_stream_t stream);
int ccnet_cca_nchw_to_planes_f32(const float *src, uint16_t *dst, int B, int C, int H, int W, long src_bs, long dst_bs,
                                 int dst_ps, int layout, ccnet_stream_t stream);
/* The module's three 1x1 projections (functions.py:29,32,35: query_conv, key_conv, value_conv; weights (Cq|Cq|C, C) fp32, biases)
 * packed for ONE stacked GEMM by ONE launch: ``w`` (2 Cq + C, C) fp32 = query | key | value rows, ``b`` (2 Cq + C) fp32, and --
 * unless both are NULL -- the bf16 operands of the split-bf16 x3 GEMMs: ``w3`` (2 Cq + C, 3 C), row n = [wh | wl | wh], and
 * ``w3t`` (C, 3 (2 Cq + C)), row c = [wh^T | wh^T | wl^T], wh = bf16_rne(w), wl = bf16_rne(w - wh).  The Python host calls it on
 * every forward instead of caching stacked weights across calls (a cache keyed on tensor versions is stale after ``p.data``
 * updates; a 3 us launch cannot be). */
int ccnet_cca_pack_projection_f32(const float *wq, const float *bq, const float *wk, const float *bk, const float *wv, const float *bv,
                                  float *w, float *b, uint16_t *w3, uint16_t *w3t, int C, int Cq, ccnet_stream_t stream);
/* The module's forward projection (functions.py:29,32,35 as ONE stacked GEMM) as a hand-written MFMA kernel (csrc/cca_gemm.hpp):
 * out[m][n] = sum_k a[m][k] * wt[n][k] + bias[n], bf16 operands, fp32 accumulation and output.  ``a`` (M, K) row stride lda -- x as
 * three bf16 planes per pixel (ccnet_cca_nchw_to_planes_f32, CCNET_PLANES_HHL: K = 3 C) --, ``wt`` (N, K) row stride ldw -- ``w3`` of
 * ccnet_cca_pack_projection_f32 --, ``out`` (M, N) row stride ldo: the pixel-major q | k | v the core reads; ``bias`` (N) or NULL
 * starts the accumulators (no epilogue pass).  K, lda, ldw % 8 == 0, ldo % 4 == 0 (elements). */
int ccnet_cca_projection_bf16(const uint16_t *a, const uint16_t *wt, const float *bias, float *out, int M, int N, int K,
                              long lda, long ldw, long ldo, ccnet_stream_t stream);
/* Its adjoint with respect to the input (the backward-data of functions.py:29,32,35), NCHW, by the same kernel in ONE launch over
 * the batch: dx[b][c][p] = sum_k w[c][k] * d[b][p][k] + add[b][c][p].  ``w`` (C, K) bf16 row stride ldw -- ``w3t`` of
 * ccnet_cca_pack_projection_f32, K = 3 (2 Cq + C) --, ``d`` (B, P, K) bf16 with pixel stride ldd and batch stride d_bs -- dq | dk | dv
 * as three planes per pixel (ccnet_cca_backward_planes3_f32) --, ``add`` (B, C, P) fp32 or NULL -- dy, the residual branch: it starts
 * the accumulators --, ``dx`` (B, C, P) fp32; add and dx contiguous, P = H W of any parity.  K, ldw, ldd, d_bs % 8 == 0. */
int ccnet_cca_projection_adjoint_bf16(const uint16_t *w, const uint16_t *d, const float *add, float *dx, int B, int C, int P, int K,
                                      long ldw, long ldd, long d_bs, ccnet_stream_t stream);
/* ... and with respect to the weight (the backward-weight of functions.py:29,32,35): sum_s part[s][n][c] = sum_r d[r][n] * x[r][c] over
 * R = 3 B H W rows -- ``d`` (R, N) bf16 row stride ldd: dq | dk | dv as three planes per pixel [dh | dl | dh], ``x`` (R, C) bf16 row
 * stride ldx: x as three planes per pixel [xh | xh | xl]; row r of one pairs with row r of the other.  The rows are cut into S slabs
 * (S x ceil(N / 128) x ceil(C / 256) workgroups: choose S so that this is about the CU count); ``part`` (S, N, C) fp32 receives one
 * partial sum per slab, every element written, and the caller adds the S partials in a fixed order (no atomics: run-to-run
 * identical).  N, C, ldd, ldx % 8 == 0. */
int ccnet_cca_projection_wgrad_bf16(const uint16_t *d, const uint16_t *x, float *part, int R, int N, int C, long ldd, long ldx, int S,
                                    ccnet_stream_t stream);
int ccnet_cca_forward_planes_f32(const float *q, const float *k, const float *v, const float *v_bias, uint16_t *v_planes,
                                 const float *x, const float *gamma, float *y, float *A,
                                 int B, int C, int Cq, int H, int W,
                                 long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long vp_bs, int vp_ps,
                                 void *workspace, size_t workspace_bytes, ccnet_stream_t stream);
/* The attention tensor alone from pixel-major q, k views (``bf16`` != 0: bf16 views as in the *_pm_bf16 entry points, else fp32
 * views as in the *_pm_f32 / *_planes_f32 ones): exactly what those forwards leave in ``A``.  The host calls it in the backward
 * pass when it did NOT keep A between forward and backward (recompute instead of save: SURVEY.md 8(f) rank 4,
 * networks/ccnet.py:118-119 -- R applications of the module hold R attention tensors). */
int ccnet_cca_attention_pm(const void *q, const void *k, float *A, int bf16, int B, int Cq, int H, int W,
                           long q_bs, int q_ps, long k_bs, int k_ps, ccnet_stream_t stream);
int ccnet_cca_backward_planes_f32(const float *dy, const float *q, const float *k, const float *v, const uint16_t *v_planes,
                                  const float *A, const float *gamma, float *dq, float *dk, float *dv, float *dgamma, float *scratch,
                                  int B, int C, int Cq, int H, int W, long q_bs, int q_ps, long k_bs, int k_ps,
                                  long v_bs, int v_ps, long vp_bs, int vp_ps, long dq_bs, int dq_ps, long dk_bs, int dk_ps, long dv_bs, int dv_ps,
                                  void *workspace, size_t workspace_bytes, ccnet_stream_t stream);
/* ccnet_cca_backward_planes_f32 (plane-free form: ``v`` fp32 pixel-major, strips <= 100, C/8 <= 64) for a caller whose NEXT
 * operation is the split-bf16 projection adjoint -- the module: dx = W^T dqkv^T and dW = dqkv^T x as bf16 -> fp32 GEMMs on
 * K-concatenated three-plane operands (functions.py:29,32,35 backwards).  dq | dk | dv are WRITTEN as the rows those GEMMs read:
 * ``d3`` = (B, HW, 3, 2 Cq + C) bf16, CCNET_PLANES_HLH (hi | lo | hi of the packed dq | dk | dv row of a pixel), pixel stride
 * ``d3_ps`` >= 3 (2 Cq + C), batch stride ``d3_bs`` (bf16 elements, both % 4 == 0); ``dbias`` (2 Cq + C floats) = the sum of that
 * row over all images and pixels (the three bias gradients), added in a fixed order.  No fp32 dq | dk | dv exists at all: the pass
 * that read it back to split it (ccnet_cca_split_planes_colsum_f32: 192 MB read + 289 MB written at (8,512,97,97)) is gone.
 * Same arithmetic as ccnet_cca_backward_planes_f32: the planes are the exact hi | lo split of its fp32 outputs (tests).
 * Workspace: CCNET_WS_PLANES3_BACKWARD.  Runs the default launch forms only (CCNET_E_BADFLAGS while an A/B option is set). */
int ccnet_cca_backward_planes3_f32(const float *dy, const float *q, const float *k, const float *v, const float *A, const float *gamma,
                                   uint16_t *d3, float *dbias, float *dgamma, float *scratch, int B, int C, int Cq, int H, int W,
                                   long q_bs, int q_ps, long k_bs, int k_ps, long v_bs, int v_ps, long d3_bs, int d3_ps,
                                   void *workspace, size_t workspace_bytes, ccnet_stream_t stream);

/* Options by name.  Both calls return a STATUS (0, or CCNET_E_BADFLAGS for an unknown name / a value outside the option's
 * range, CCNET_E_NULLPTR); values travel through out-parameters (``previous`` may be NULL), so that an option value of -1 is
 * never mistaken for an error.  Defaults are what ships:
 *   "impl" CCNET_IMPL_*, "precision" CCNET_PRECISION_*, "branch_mask" CCNET_BRANCH_* (see above);
 * development / A-B switches:
 *   "planes_ring"  which kernels run the split-plane passes that have a pixel-major output:
 *                  2 (default) gmap3_kernel -- stores straight from the accumulators; column passes with two ring slots and
 *                    three workgroups per CU, row passes with three ring slots and two workgroups per CU;
 *                  1 gmap3_kernel, three ring slots / two workgroups per CU everywhere;
 *                  0 gmap_kernel (two feature tiles + an output image in LDS).
 *   "planes_stream" 1 (default) the split-plane dA contraction is the persistent gweight_stream_kernel (one workgroup per CU
 *                    walks the strips, its three-stage ring runs across strip boundaries); 0 gweight_kernel (one workgroup
 *                    per strip); k > 1: persistent with at most k workgroups (tests).
 *   "planes_overlap" the backwards of the pixel-major / split-plane entries run their two dv passes on a library-owned side
 *                    stream (forked from ``stream`` by an event, joined before the call returns: the caller's stream sees
 *                    one ordered operation and a stream capture stays one graph): 2 next to dA, softmax-backward and
 *                    dq | dk; 1 next to softmax-backward and dq | dk only; 0 everything on ``stream``; -1 (default) what
 *                    measured best per family (2 on split planes, 1 on the bf16 / fp32 pixel-major entries).
 *   "planes_xcd"   1 (default): the NCHW row pass of the split-plane forward decodes its strips XCD-aware (consecutive rows of an
 *                    image on one XCD, whose L2 then merges the boundary lines neighbouring NCHW rows share); 0: linear.
 *   "da_stages"    2 (default) / 3: LDS ring stages of the persistent dA kernel of the plane-free backward.  Three fill the CU's LDS,
 *                    so the dv column pass on the side stream waits for its workgroups to exit; two leave room for one column
 *                    workgroup per CU and the launches overlap for real (backward 0.45 -> 0.40 ms at the headline shape).
 *   "dqdk_wpc3"    1 (default): ca_backward of the fp32 pixel-major / split-plane entries at strips <= 100 and C/8 <= 64 (one channel
 *                    group per strip) runs the one-slot form of its kernel: 53.6 KB of LDS, <= 168 VGPRs, three workgroups per CU
 *                    instead of two (these launches are latency chains); 0: the two-slot form.  Same arithmetic, same bits.
 *   "energy_tail"  1 (default): the fp32 energies launch of the pixel-major / split-plane entries (strips <= 100, C/8 <= 64) cuts the
 *                    strips beyond its whole rounds of workgroups into tile-row parts (a short last round); 0: one workgroup per strip.
 *   "bf16_partial" 1 (default): in the bf16 pixel-major entry points the column -> row partial of the aggregation and of dv is a bf16
 *                    tensor (the reference's own bf16 arithmetic rounds out_H and the column half of dv to bf16 before adding the row
 *                    half, functions.py:46-47): half the bytes of the fp32 partial, which was a quarter of the traffic of BASELINE
 *                    configs[4]; 0: fp32 partial (round 2-4 behaviour).  fp32 families are not affected.
 *   "dqdk_exact"   ca_backward (dq | dk) of EVERY fp32 pixel-major / split-plane route, strips of any supported length:
 *                    1 (default) six bf16 terms of a three-way split per product -- fp32-equivalent (2^-24) at 3/8 of the matrix time
 *                    of v_mfma_f32_16x16x4_f32, +3..4 us per launch at (8,512,97,97) over the three-term form; what is left on dq | dk is
 *                    what the upstream dA carries (~5e-6 of max |dq|).  0 = three terms (~1.2e-5 of max |dq|, |dk|: 4e-4 at the
 *                    reference's initialisation scale, 1.5e-3 with q, k three times hotter); strips <= 100 only, an A/B aid. */
int ccnet_cca_set_option(const char *name, int value, int *previous);
int ccnet_cca_get_option(const char *name, int *value);

/* Launch profiler (a measurement aid, off by default).  Between ``begin`` and ``end`` every kernel launch the library
 * issues is bracketed by a HIP-event pair on its stream; ``end`` disarms, waits for the recorded launches and returns
 * their count (>= 0), filling ms[i] with launch i's duration and names + i * name_stride with its kernel name.
 * Not re-entrant (one armed session per process); negative return = error.  Do not arm during a stream capture. */
int ccnet_cca_profile_begin(int max_launches);
int ccnet_cca_profile_end(float *ms, char *names, int name_stride, int cap);

/* Device-state probes (measurement aids; bench.py prints what they read on the metric's line, see csrc/cca_probe.hpp).
 * A wave reads two counters: shader cycles and the constant 100 MHz reference clock -- the ratio of two deltas is the clock
 * it really ran at, which rocm-smi's requested level does not show under a power / thermal limit.
 * ccnet_cca_probe_clock: ``nwg`` single-wave workgroups sample (shader, reference) every ``interval_ticks`` reference ticks,
 *   ``nsamples`` times: samples[(wg * nsamples + s) * 2 + {0, 1}], then samples[2 * nwg * nsamples + wg] = the XCC of workgroup
 *   wg (buffer: (2 * nwg * nsamples + nwg) uint64).  Launch it on a stream of its own NEXT TO what is to be observed.
 *   nsamples * interval_ticks <= 2e8 (two seconds of busy-waiting), otherwise CCNET_E_BADFLAGS.
 * ccnet_cca_probe_mfma: ``nwg`` workgroups of 4 waves, each wave ``iters`` x 8 v_mfma_f32_16x16x32_bf16 (131072 * iters flops);
 *   clk[wg * 4 + {0..3}] = shader start, reference start, shader end, reference end of wave 0; ``sink``: 256 * nwg floats.
 * ccnet_cca_probe_dma: ``nwg`` workgroups fill a 25 KiB LDS tile by LDS-DMA (25 pieces of 4 rows x 256 B, rows
 *   ``row_stride_bytes`` apart: a column strip of a pixel-major tensor) ``reps`` times from random places of ``src``;
 *   clk[wg * 4 + {0..3}] = sum of the issue -> landed latencies (shader cycles), their maximum, reference start / end. */
int ccnet_cca_probe_clock(unsigned long long *samples, int nwg, int nsamples, int interval_ticks, ccnet_stream_t stream);
int ccnet_cca_probe_mfma(unsigned long long *clk, float *sink, int nwg, int iters, ccnet_stream_t stream);
int ccnet_cca_probe_dma(const float *src, size_t src_bytes, unsigned long long *clk, int nwg, int reps, int row_stride_bytes,
                        ccnet_stream_t stream);

/* Device self-test of the MFMA fragment layout the kernels assume (asymmetric operands).
 * ``scratch`` >= 64 bytes of device memory.  The one entry point that synchronises ``stream`` (it copies
 * the verdict back).  Returns 0 when the layout matches, 1000 + #mismatches otherwise. */
int ccnet_cca_mfma_selftest(float *scratch, ccnet_stream_t stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* CCNET_CCA_H */
