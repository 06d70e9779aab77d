/*
 * oracle/msda_oracle_impl.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's multi-scale deformable attention
 * arithmetic.  Included twice by msda_oracle.c (REAL = float, REAL = double).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this; the product path (monodetr_amd/) never links or imports it.
 *
 * Follows (reference paths relative to /root/reference/lib/models/monodetr/ops):
 *   forward   src/cuda/ms_deform_im2col_cuda.cuh:237-299 (per-output loop)
 *             src/cuda/ms_deform_im2col_cuda.cuh:33-84   (bilinear gather)
 *   backward  src/cuda/ms_deform_im2col_cuda.cuh:301-403 (per-sample loop,
 *             channel reduction started at c=0 and summed in c order :377-393)
 *             src/cuda/ms_deform_im2col_cuda.cuh:87-159  (bilinear scatter)
 *   shapes    src/cuda/ms_deform_attn_cuda.cu:20-153
 *
 * Pixel coordinate: the reference writes `loc_h * spatial_h - 0.5` with a
 * *double* literal (.cuh:285-286), so for float the product is rounded to
 * float first and the subtraction cannot be contracted into an FMA.  This
 * file is compiled with -ffp-contract=off so REAL_MUL / REAL_SUB round
 * separately, which is what "bit-exact gather indices" is defined against.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* One bilinear footprint: floor indices, fractional weights, corner validity.
 * .cuh:38-45 (indices/weights), :56,:62,:68,:74 (corner bounds). */
typedef struct {
    int h_low, w_low;
    REAL w1, w2, w3, w4;   /* hh*hw, hh*lw, lh*hw, lh*lw            (.cuh:80) */
    REAL lh, lw, hh, hw;
    int ok1, ok2, ok3, ok4; /* (low,low) (low,high) (high,low) (high,high)     */
} FN(footprint);

static inline int FN(in_window)(REAL h_im, REAL w_im, int H, int W)
{
    /* .cuh:288 */
    return h_im > (REAL)-1 && w_im > (REAL)-1 && h_im < (REAL)H && w_im < (REAL)W;
}

/* cell: NULL = the sample's own cell (floor, .cuh:38-39); otherwise (h_low, w_low) GIVEN by the caller -- the "forced cell"
 * evaluation used by tests/test_model_gpu.py: the float64 model is evaluated on the bilinear patch the fp32 run chose for each
 * sample (the output is continuous across a cell boundary, its location derivative is not; a sample within fp32 resolution of a
 * boundary may floor differently in the two precisions).  The weights lh / lw are then the same polynomial extended past [0, 1]. */
static inline void FN(make_footprint)(REAL h, REAL w, int H, int W, FN(footprint) *f, const int32_t *cell)
{
    f->h_low = cell ? (int)cell[0] : (int)FLOOR(h);
    f->w_low = cell ? (int)cell[1] : (int)FLOOR(w);
    const int h_high = f->h_low + 1, w_high = f->w_low + 1;
    f->lh = h - (REAL)f->h_low;
    f->lw = w - (REAL)f->w_low;
    f->hh = (REAL)1 - f->lh;
    f->hw = (REAL)1 - f->lw;
    f->w1 = f->hh * f->hw; f->w2 = f->hh * f->lw;
    f->w3 = f->lh * f->hw; f->w4 = f->lh * f->lw;
    f->ok1 = f->h_low >= 0 && f->w_low >= 0;
    f->ok2 = f->h_low >= 0 && w_high <= W - 1;
    f->ok3 = h_high <= H - 1 && f->w_low >= 0;
    f->ok4 = h_high <= H - 1 && w_high <= W - 1;
}

/* value [B,S,M,D]; shapes [L,2] (H,W); level_start [L]; loc [B,Lq,M,L,P,2] (x,y);
 * attn [B,Lq,M,L,P]; out [B,Lq,M*D].  Returns 0. */
/* forced: NULL, or int32 [B,Lq,M,L,P,4] = (in_window, h_low, w_low, -) as msda_oracle_indices writes it: window test and cell
 * of every sample taken from there instead of from this precision's own arithmetic. */
static int FN(msda_forward_impl)(const REAL *value, const int64_t *shapes, const int64_t *level_start,
                            const REAL *loc, const REAL *attn, REAL *out,
                            int B, int S, int M, int D, int L, int Lq, int P, const int32_t *forced)
{
    const int64_t qid_stride = (int64_t)M * D;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int q = 0; q < Lq; ++q) {
            for (int m = 0; m < M; ++m) {
                const int64_t samp = ((int64_t)b * Lq + q) * M + m;   /* .cuh:258 sampling_index */
                REAL *col = out + samp * D;
                for (int c = 0; c < D; ++c) col[c] = (REAL)0;
                const REAL *locp = loc + samp * L * P * 2;
                const REAL *attp = attn + samp * L * P;
                for (int l = 0; l < L; ++l) {
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                    const REAL *vbase = value + ((int64_t)b * S + level_start[l]) * qid_stride;
                    for (int p = 0; p < P; ++p) {
                        const REAL loc_w = locp[(l * P + p) * 2];
                        const REAL loc_h = locp[(l * P + p) * 2 + 1];
                        const REAL weight = attp[l * P + p];
                        const REAL h_im = loc_h * (REAL)H - (REAL)0.5;   /* .cuh:285 */
                        const REAL w_im = loc_w * (REAL)W - (REAL)0.5;   /* .cuh:286 */
                        const int32_t *fc = forced ? forced + ((samp * L + l) * P + p) * 4 : NULL;
                        if (fc ? !fc[0] : !FN(in_window)(h_im, w_im, H, W)) continue;
                        FN(footprint) f;
                        FN(make_footprint)(h_im, w_im, H, W, &f, fc ? fc + 1 : NULL);
                        const int64_t o_ll = ((int64_t)f.h_low * W + f.w_low) * qid_stride + (int64_t)m * D;
                        const int64_t dx = qid_stride, dy = (int64_t)W * qid_stride;
                        for (int c = 0; c < D; ++c) {
                            const REAL v1 = f.ok1 ? vbase[o_ll + c] : (REAL)0;
                            const REAL v2 = f.ok2 ? vbase[o_ll + dx + c] : (REAL)0;
                            const REAL v3 = f.ok3 ? vbase[o_ll + dy + c] : (REAL)0;
                            const REAL v4 = f.ok4 ? vbase[o_ll + dy + dx + c] : (REAL)0;
                            const REAL val = f.w1 * v1 + f.w2 * v2 + f.w3 * v3 + f.w4 * v4;  /* .cuh:82 */
                            col[c] += val * weight;                                         /* .cuh:290 */
                        }
                    }
                }
            }
        }
    }
    return 0;
}

int FN(msda_oracle_forward)(const REAL *value, const int64_t *shapes, const int64_t *level_start,
                            const REAL *loc, const REAL *attn, REAL *out,
                            int B, int S, int M, int D, int L, int Lq, int P)
{
    return FN(msda_forward_impl)(value, shapes, level_start, loc, attn, out, B, S, M, D, L, Lq, P, NULL);
}

int FN(msda_oracle_forward_forced)(const REAL *value, const int64_t *shapes, const int64_t *level_start,
                                   const REAL *loc, const REAL *attn, REAL *out, const int32_t *forced,
                                   int B, int S, int M, int D, int L, int Lq, int P)
{
    return FN(msda_forward_impl)(value, shapes, level_start, loc, attn, out, B, S, M, D, L, Lq, P, forced);
}

/* grad_out [B,Lq,M*D] -> grad_value [B,S,M,D], grad_loc (shape of loc), grad_attn (shape of attn).
 * All three outputs are fully (re)written; no pre-zeroing needed.  Parallel over (image, head):
 * heads write disjoint channels of grad_value, and inside one (image, head) the scatter stays
 * sequential in (q,l,p,c) order -- every address sees its contributions in the same order as a
 * fully serial (q,m,l,p,c) loop, so the oracle is deterministic and thread-count independent
 * (the reference's atomics are not). */
static int FN(msda_backward_impl)(const REAL *value, const int64_t *shapes, const int64_t *level_start,
                             const REAL *loc, const REAL *attn, const REAL *grad_out,
                             REAL *grad_value, REAL *grad_loc, REAL *grad_attn,
                             int B, int S, int M, int D, int L, int Lq, int P, const int32_t *forced)
{
    const int64_t qid_stride = (int64_t)M * D;
    memset(grad_value, 0, sizeof(REAL) * (size_t)B * S * M * D);
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int m = 0; m < M; ++m) {
            for (int q = 0; q < Lq; ++q) {
                const int64_t samp = ((int64_t)b * Lq + q) * M + m;
                const REAL *g = grad_out + samp * D;
                const REAL *locp = loc + samp * L * P * 2;
                const REAL *attp = attn + samp * L * P;
                REAL *glocp = grad_loc + samp * L * P * 2;
                REAL *gattp = grad_attn + samp * L * P;
                for (int l = 0; l < L; ++l) {
                    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
                    const int64_t voff = ((int64_t)b * S + level_start[l]) * qid_stride;
                    const REAL *vbase = value + voff;
                    REAL *gvbase = grad_value + voff;
                    for (int p = 0; p < P; ++p) {
                        const REAL loc_w = locp[(l * P + p) * 2];
                        const REAL loc_h = locp[(l * P + p) * 2 + 1];
                        const REAL weight = attp[l * P + p];
                        const REAL h_im = loc_h * (REAL)H - (REAL)0.5;
                        const REAL w_im = loc_w * (REAL)W - (REAL)0.5;
                        REAL acc_w = (REAL)0, acc_h = (REAL)0, acc_a = (REAL)0;  /* .cuh:365-367 */
                        const int32_t *fc = forced ? forced + ((samp * L + l) * P + p) * 4 : NULL;
                        if (fc ? fc[0] : FN(in_window)(h_im, w_im, H, W)) {
                            FN(footprint) f;
                            FN(make_footprint)(h_im, w_im, H, W, &f, fc ? fc + 1 : NULL);
                            const int64_t o_ll = ((int64_t)f.h_low * W + f.w_low) * qid_stride + (int64_t)m * D;
                            const int64_t dx = qid_stride, dy = (int64_t)W * qid_stride;
                            for (int c = 0; c < D; ++c) {
                                const REAL top_grad = g[c];
                                const REAL tgv = top_grad * weight;                 /* .cuh:113 */
                                REAL gh = (REAL)0, gw = (REAL)0;
                                REAL v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                                if (f.ok1) { v1 = vbase[o_ll + c];           gh -= f.hw * v1; gw -= f.hh * v1; gvbase[o_ll + c]           += f.w1 * tgv; }
                                if (f.ok2) { v2 = vbase[o_ll + dx + c];      gh -= f.lw * v2; gw += f.hh * v2; gvbase[o_ll + dx + c]      += f.w2 * tgv; }
                                if (f.ok3) { v3 = vbase[o_ll + dy + c];      gh += f.hw * v3; gw -= f.lh * v3; gvbase[o_ll + dy + c]      += f.w3 * tgv; }
                                if (f.ok4) { v4 = vbase[o_ll + dy + dx + c]; gh += f.lw * v4; gw += f.lh * v4; gvbase[o_ll + dy + dx + c] += f.w4 * tgv; }
                                const REAL val = f.w1 * v1 + f.w2 * v2 + f.w3 * v3 + f.w4 * v4;
                                acc_a += top_grad * val;                             /* .cuh:156 */
                                acc_w += (REAL)W * gw * tgv;                         /* .cuh:157 */
                                acc_h += (REAL)H * gh * tgv;                         /* .cuh:158 */
                            }
                        }
                        glocp[(l * P + p) * 2] = acc_w;
                        glocp[(l * P + p) * 2 + 1] = acc_h;
                        gattp[l * P + p] = acc_a;
                    }
                }
            }
        }
    }
    return 0;
}

int FN(msda_oracle_backward)(const REAL *value, const int64_t *shapes, const int64_t *level_start,
                             const REAL *loc, const REAL *attn, const REAL *grad_out,
                             REAL *grad_value, REAL *grad_loc, REAL *grad_attn,
                             int B, int S, int M, int D, int L, int Lq, int P)
{
    return FN(msda_backward_impl)(value, shapes, level_start, loc, attn, grad_out, grad_value, grad_loc, grad_attn,
                                  B, S, M, D, L, Lq, P, NULL);
}

int FN(msda_oracle_backward_forced)(const REAL *value, const int64_t *shapes, const int64_t *level_start,
                                    const REAL *loc, const REAL *attn, const REAL *grad_out,
                                    REAL *grad_value, REAL *grad_loc, REAL *grad_attn, const int32_t *forced,
                                    int B, int S, int M, int D, int L, int Lq, int P)
{
    return FN(msda_backward_impl)(value, shapes, level_start, loc, attn, grad_out, grad_value, grad_loc, grad_attn,
                                  B, S, M, D, L, Lq, P, forced);
}

/* Gather indices per sample, for the "bit-exact index" parity check.
 * idx [B,Lq,M,L,P,4] int32 = (in_window, h_low, w_low, corner_mask); h_low/w_low/corner_mask
 * are 0 when the sample is outside the window (.cuh:288). corner_mask bit k = ok(k+1). */
int FN(msda_oracle_indices)(const int64_t *shapes, const REAL *loc, int32_t *idx,
                            int B, int M, int L, int Lq, int P)
{
    const int64_t n = (int64_t)B * Lq * M;
    for (int64_t s = 0; s < n; ++s) {
        for (int l = 0; l < L; ++l) {
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
            for (int p = 0; p < P; ++p) {
                const int64_t k = (s * L + l) * P + p;
                const REAL h_im = loc[k * 2 + 1] * (REAL)H - (REAL)0.5;
                const REAL w_im = loc[k * 2] * (REAL)W - (REAL)0.5;
                int32_t *o = idx + k * 4;
                o[0] = o[1] = o[2] = o[3] = 0;
                if (FN(in_window)(h_im, w_im, H, W)) {
                    FN(footprint) f;
                    FN(make_footprint)(h_im, w_im, H, W, &f, NULL);
                    o[0] = 1; o[1] = f.h_low; o[2] = f.w_low;
                    o[3] = f.ok1 | (f.ok2 << 1) | (f.ok3 << 2) | (f.ok4 << 3);
                }
            }
        }
    }
    return 0;
}

#undef FN
#undef CAT
#undef CAT_
