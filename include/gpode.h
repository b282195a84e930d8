/* gpode.h -- C ABI of the MI355X-native GP-ODE hot path (libgpode_hip.so).
 *
 * Drop-in boundary for the reference's Python operator API
 * (/root/reference/experiments/model/core/{svpy,kernels,flow}.py).  The reference has no FFI
 * layer of its own; each entry point below names the reference method it replaces, and
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to contiguous row-major fp32 unless stated otherwise;
 *  - the library never allocates, frees or retains caller memory; scratch is passed in (`ws`);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *  - return value 0 = ok, non-zero = error (message via gpode_last_error()); the Python host
 *    raises RuntimeError on non-zero;
 *  - all randomness enters as explicit tensors: eps_u (M,Do), rff_w (S,Do) [DF (2S,Do)],
 *    rff_eps (Di,S,Do), rff_u (1,S,Do) in [0,1).
 *
 *  kernel: 0 = RBF dimwise  (kernels.py:29-195),  1 = divergence-free (kernels.py:201-393, Di==Do)
 *  order : 1 | 2            (flow.py:27-45)
 *  method: 0 = euler, 1 = rk4 (torchdiffeq fixed-grid 3/8 rule; flow.py:76-85), 2 = midpoint (y1 = y + dt f(y + dt/2 f(y)))
 */
#ifndef GPODE_H
#define GPODE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GPODE_KERNEL_RBF 0
#define GPODE_KERNEL_DF 1
#define GPODE_METHOD_EULER 0
#define GPODE_METHOD_RK4 1

/* Library / build identification. */
const char* gpode_version(void);
const char* gpode_last_error(void);
/* 1 if (kernel,Di,Do) has a compiled specialisation. */
int gpode_supported(int kernel, int Di, int Do);

/* Sizes (in floats) of the packed per-draw cache consumed by rhs/rollout, and of the scratch
 * workspace gpode_cache_build_fwd needs. */
int gpode_cache_sizes(int kernel, int Di, int Do, int M, int S, size_t* pack_floats, size_t* ws_floats);

/* SVGP_Layer.build_cache (svpy.py:103-121): kern.build_cache (kernels.py:126-137 / :305-316),
 * sample_inducing (svpy.py:88-101), kern.K(Z) (kernels.py:98-110 / :289-303), kern.rff_forward(Z),
 * kern.compute_nu (kernels.py:155-172 / :376-387).
 * Inputs are the raw optvars of the state_dict and the noise.  `pack` receives the lane-major cache
 * used by the kernels below.  The remaining outputs mirror the attributes the reference caches on
 * `kern` and may be NULL: ell (Do,Di), var (Do), omega (Di,S,Do), phase (1,S,Do), u (M,Do),
 * Lu (RBF: (Do,M,M); DF: (M*D,M*D)), nu (RBF: (Do,M); DF: (M*D)), u_prior (M,Do). */
int gpode_cache_build_fwd(int kernel, int Di, int Do, int M, int S,
                          const float* raw_ell, const float* raw_var, const float* Z,
                          const float* Um, const float* Us_packed,
                          const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                          float* pack, float* ws,
                          float* ell, float* var, float* omega, float* phase, float* u,
                          float* Lu, float* nu, float* u_prior, void* stream);

/* SVGP_Layer.forward (svpy.py:123-142): f(x) = rff_forward(x) + f_update(x, Z).
 * x (N,Di) -> f (N,Do).  mode: 0 = prior + update, 1 = prior only, 2 = update only. */
int gpode_rhs_fwd(int kernel, int Di, int Do, int M, int S, const float* pack,
                  const float* x, int N, float* f, int mode, void* stream);

/* Flow.forward (flow.py:68-86) given a built cache: z0 (N,D), ts (T) -> zt (N,T,D),
 * D = Di = order*Do.  One fixed-grid step per output interval.
 * xstage (nullable): (N, T-1, NS, D) receives the input of every RHS evaluation (NS = 1 euler, 4 rk4);
 * pass it when gradients are needed -- gpode_rollout_bwd walks it backwards. */
int gpode_rollout_fwd(int kernel, int order, int method, int Di, int Do, int M, int S,
                      const float* pack, const float* z0, const float* ts, int N, int T,
                      float* zt, float* xstage, void* stream);

/* Gradient of a scalar loss through Flow.forward = what loss.backward() computes through the unrolled
 * solver in the reference (main.py:209-210, use_adjoint=False).
 *   gzt (N,T,D) = dL/dzt  ->  gz0 (N,D) = dL/dz0  and  astage (N,T-1,NS,Do) = dL/df at every RHS evaluation. */
int gpode_rollout_bwd(int kernel, int order, int method, int Di, int Do, int M, int S,
                      const float* pack, const float* xstage, const float* gzt, const float* ts, int N, int T,
                      float* gz0, float* astage, void* stream);

/* gx (R,Di) = J_f(x)^T a for rows x (R,Di), a (R,Do): autograd of SVGP_Layer.forward w.r.t. its input. */
int gpode_rhs_vjp(int kernel, int Di, int Do, int M, int S, const float* pack,
                  const float* x, const float* a, int R, float* gx, void* stream);

/* Parameter gradient of sum_r <a_r, f(x_r)> in PACK LAYOUT (same indexing as `pack`: d/d of every record
 * field and of the uniform tail); rows x (R,Di), a (R,Do).  slab: nchunk * pack_floats floats of scratch.
 * accumulate != 0 adds into gpack.  Deterministic (fixed-order two-stage sum, no float atomics). */
int gpode_param_grad(int kernel, int Di, int Do, int M, int S, const float* pack,
                     const float* x, const float* a, int R, float* slab, int nchunk, float* gpack, int accumulate,
                     void* stream);

/* Backward of gpode_cache_build_fwd: what autograd does behind SVGP_Layer.build_cache in the reference
 * (nu = L^-T(u - L^-1 f_prior(Z)), cholesky, K(Z), omega = eps/ell, softplus ...).
 *   gpack : in/out, pack-layout gradient from gpode_param_grad (the f_prior(Z) path is added to it);
 *   ws    : the workspace the forward of THIS draw wrote;  bws: gpode_cache_bwd_sizes() floats of scratch;
 *   outputs: gradients w.r.t. the five raw parameter tensors, in their state_dict layouts;
 *   prepared: bit 0 -- bws went through gpode_cache_bwd_prepare (below); bit 1 -- g_Um / g_Us already HOLD a gradient (that of
 *   KL(q(u)||p(u)), svpy.py:144-175, which reaches the same two parameters) and the flow's is ADDED to it: what autograd's
 *   accumulation of the two paths would do in one more launch each. */
int gpode_cache_bwd_sizes(int kernel, int Di, int Do, int M, int S, size_t* bws_floats);
int gpode_cache_build_bwd(int kernel, int Di, int Do, int M, int S,
                          const float* raw_ell, const float* raw_var, const float* Z, const float* eps_u,
                          const float* pack, const float* ws, float* gpack, float* bws,
                          float* g_raw_ell, float* g_raw_var, float* g_Z, float* g_Um, float* g_Us, int prepared, void* stream);
/* The gradient-independent part of the above (L^-1 from the factor in ws, written into bws).  Optional: call it any time
 * after gpode_cache_build_fwd -- e.g. on a side stream while the decoder runs -- and pass prepared = 1 with the same bws. */
int gpode_cache_bwd_prepare(int kernel, int Di, int Do, int M, int S, const float* ws, float* bws, void* stream);

/* ---- Monte-Carlo draws batched into one call ------------------------------------------------------------------------------
 * ODEGPVAE.sample_trajectories (odegpvae.py:37-45) loops `for l in range(L)` over whole flow calls -- L cache builds, L solver
 * loops, L autograd graphs -- and the training loop runs half of all epochs at L = 5 (main.py:200).  K_uu + jitter I and its
 * Cholesky factor depend on the parameters only (kernels.py:163 / :384), so the `_n` forms below build ONE factor for all
 * `ndraws` function draws (their right-hand sides f_prior_l(Z) are solved as a block), integrate the L * N trajectories in ONE
 * launch (the launch's second grid dimension is the draw: its own pack, the shared initial states) and run ONE Cholesky
 * backward on the summed adjoint (that VJP is linear).  Layouts: every per-draw operand gains a LEADING draw axis --
 *   eps_u (L,M,Do), rff_w (L,S|2S,Do), rff_eps (L,Di,S,Do), rff_u (L,1,S,Do); pack (L, pack_floats); omega / phase / u / nu /
 *   u_prior (L, ...); zt (L,N,T,D), xstage (L,N,T-1,NS,D), gzt, gz0 (L,N,D), astage; x / a of gpode_param_grad_n (L,R,.),
 *   slab (L, nchunk, pack_floats), gpack (L, pack_floats) --
 * while z0 (N,D), ts, the raw parameters, ell, var, Lu and the factor inside `ws` are shared.  gpode_cache_build_bwd_n returns the
 * parameter gradients SUMMED over the draws (what autograd accumulates over the reference's loop).  ws / bws sizes depend on
 * ndraws (gpode_cache_sizes_n, gpode_cache_bwd_sizes_n); pack_floats does not.  ndraws = 1 is the un-suffixed entry point. */
int gpode_cache_sizes_n(int kernel, int Di, int Do, int M, int S, int ndraws, size_t* pack_floats, size_t* ws_floats);
int gpode_cache_build_fwd_n(int kernel, int Di, int Do, int M, int S, int ndraws,
                            const float* raw_ell, const float* raw_var, const float* Z,
                            const float* Um, const float* Us_packed,
                            const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                            float* pack, float* ws,
                            float* ell, float* var, float* omega, float* phase, float* u,
                            float* Lu, float* nu, float* u_prior, void* stream);
int gpode_rollout_fwd_n(int kernel, int order, int method, int Di, int Do, int M, int S, int ndraws,
                        const float* pack, const float* z0, const float* ts, int N, int T,
                        float* zt, float* xstage, void* stream);
int gpode_rollout_bwd_n(int kernel, int order, int method, int Di, int Do, int M, int S, int ndraws,
                        const float* pack, const float* xstage, const float* gzt, const float* ts, int N, int T,
                        float* gz0, float* astage, void* stream);
/* gpode_rollout_bwd_n and gpode_param_grad_n in ONE pass: the reverse sweep visits every (stage input, adjoint) row anyway, so the
 * rows' parameter-gradient terms are accumulated on the way and come out as gpack (ndraws, pack_floats) -- what the two calls
 * produce together, with one launch and one pass over the rows less.  slab: ndraws * nchunk * pack_floats floats of scratch with
 * nchunk = gpode_rollout_bwd_pgrad_chunks(...) -- 0 means this shape has no fused form (streamed teams, second-order models,
 * midpoint rule): call the two entry points then.  gz0 / astage as gpode_rollout_bwd_n. */
int gpode_rollout_bwd_pgrad_chunks(int kernel, int order, int method, int Di, int Do, int M, int S, int N);
int gpode_rollout_bwd_pgrad_n(int kernel, int order, int method, int Di, int Do, int M, int S, int ndraws,
                              const float* pack, const float* xstage, const float* gzt, const float* ts, int N, int T,
                              float* gz0, float* astage, float* slab, int nchunk, float* gpack, void* stream);
int gpode_param_grad_n(int kernel, int Di, int Do, int M, int S, int ndraws, const float* pack,
                       const float* x, const float* a, int R, float* slab, int nchunk, float* gpack, int accumulate,
                       void* stream);
int gpode_cache_bwd_sizes_n(int kernel, int Di, int Do, int M, int S, int ndraws, size_t* bws_floats);
int gpode_cache_build_bwd_n(int kernel, int Di, int Do, int M, int S, int ndraws,
                            const float* raw_ell, const float* raw_var, const float* Z, const float* eps_u,
                            const float* pack, const float* ws, float* gpack, float* bws,
                            float* g_raw_ell, float* g_raw_var, float* g_Z, float* g_Um, float* g_Us, int prepared, void* stream);
int gpode_cache_bwd_prepare_n(int kernel, int Di, int Do, int M, int S, int ndraws, const float* ws, float* bws, void* stream);

/* Factorisation status of the last gpode_cache_build_fwd on `ws` (bit 0: K_uu + jitter I not positive
 * definite -- torch.linalg.cholesky raises there, kernels.py:163/:384).  Copies one int to the host and
 * synchronises `stream`. */
int gpode_cache_info(const float* ws, int* host_info, void* stream);
/* Conditioning estimate of the draw's factor(s): host_min_max[0] / [1] = the smallest / largest diagonal entry of L
 * (K_uu + jitter I = L L^T, kernels.py:163 / :384); their ratio squared bounds cond(K_uu) from below.  Synchronises `stream`. */
int gpode_cache_pivots(const float* ws, float* host_min_max, void* stream);
/* Route of gpode_cache_build_bwd through the factor (the autograd of kernels.py:163-171 / :384-386): 0 auto (triangular solves up
 * to 192 rows, an explicit triangular inverse beyond), 1 always triangular solves (up to 1216 rows; slower, and as accurate as
 * torch's fp32 solves on a numerically rank-deficient K_uu, where the explicit inverse is not), 2 never.  Process-wide. */
int gpode_set_backward_solves(int mode);

/* kern.K(X, X2) (kernels.py:98-110 / :289-303), no jitter.  X (N,Di), X2 (M2,Di).
 * RBF: out (Do,N,M2).  DF: out (N*D, M2*D), row (n,a), col (m,b). */
int gpode_kernel_matrix(int kernel, int Di, int Do, const float* raw_ell, const float* raw_var,
                        const float* X, int N, const float* X2, int M2, float* out, void* stream);

/* The kernel's own methods, for a caller that keeps its SVGP_Layer class and binds method by method -- the stages of
 * gpode_cache_build_fwd, cut where the reference cuts them:
 *   gpode_kern_cache   kern.build_cache(S, device) (kernels.py:126-137 / :305-316) with kern.sample_freq (kernels.py:112-124) inside:
 *                      omega (Di,S,Do) = rff_eps / ell^T, phase (1,S,Do) = 2 pi rff_u (either may be NULL), and in `pack` a
 *                      PRIOR-ONLY cache (no inducing records) that gpode_rhs_fwd(..., M = 0, ..., mode = 1) evaluates:
 *                      kern.rff_forward(x, S) (kernels.py:140-153 / :319-351) on exactly these Fourier features.
 *   gpode_compute_nu   kern.compute_nu(Ku, u_prior, inducing_val) (kernels.py:155-172 / :376-387): Cholesky of the CALLER's
 *                      Ku + 1e-5 I (RBF (Do,M,M); DF (M D, M D)) and nu = L^-T (u - L^-1 u_prior); u_prior, u (M,Do); nu RBF (Do,M,1),
 *                      DF (M D,1).  ws: gpode_compute_nu_ws floats; afterwards gpode_cache_info(ws) reports a non-positive-definite Ku.
 *   gpode_f_update     kern.f_update(x, x2) (kernels.py:174-181 / :390-393): out (N,Do) = K(x, x2) nu for the caller's nu and x2 (M,Di).
 * `pack` / `scratch`: gpode_kern_scratch(kernel, Di, Do, M, S) floats (M = 0 for gpode_kern_cache, S = 0 for gpode_f_update); 0 = no
 * specialisation. */
size_t gpode_kern_scratch(int kernel, int Di, int Do, int M, int S);
int gpode_kern_cache(int kernel, int Di, int Do, int S, const float* raw_ell, const float* raw_var, const float* rff_w,
                     const float* rff_eps, const float* rff_u, float* pack, float* omega, float* phase, void* stream);
int gpode_compute_nu_ws(int kernel, int Di, int Do, int M, size_t* ws_floats);
int gpode_compute_nu(int kernel, int Di, int Do, int M, const float* Ku, const float* u_prior, const float* u, float* nu, float* ws,
                     void* stream);
int gpode_f_update(int kernel, int Di, int Do, int M, const float* raw_ell, const float* raw_var, const float* x2, const float* nu,
                   const float* x, int N, float* out, float* scratch, void* stream);

/* SVGP_Layer.build_conditional (svpy.py:176-210), RBF kernel: q(f(x)) = N(m(x), Sigma(x)) at the rows of x (N,Di).
 * raw_ell (Do,Di), raw_var (Do), Z (M,Di), Um (M,Do) as in gpode_cache_build_fwd.  us_rank1 = 0: Us = the packed lower
 * triangle (Do, M(M+1)/2); us_rank1 = 1 (q_diag=True): Us = the constrained scale as (Do,M) columns s_d, and Us Us^T is the
 * rank-one s_d s_d^T the reference forms from its (M,1) column (svpy.py:194-195).
 * mean (N,Do).  full_cov = 0: var (N,Do) marginal variances; full_cov = 1: var (N,N,Do) (the layout `var.T` has in
 * svpy.py:210).  ws: gpode_conditional_ws floats of scratch (the caller allocates). */
int gpode_conditional_ws(int Di, int Do, int M, int N, size_t* ws_floats);
int gpode_conditional(int Di, int Do, int M, int N, const float* raw_ell, const float* raw_var, const float* Z,
                      const float* Um, const float* Us, int us_rank1, const float* x, int full_cov, float* mean,
                      float* var, float* ws, void* stream);

/* SVGP_Layer.kl (svpy.py:144-175, q_diag=False): Um (M,Do), Us_packed (Do, M(M+1)/2) -> kl (1 float).
 * _bwd: g = d loss / d kl (1 float, device) -> dUm (M,Do), dUs (Do, M(M+1)/2). */
int gpode_svgp_kl_fwd(int M, int Do, const float* Um, const float* Us_packed, float* kl, void* stream);
int gpode_svgp_kl_bwd(int M, int Do, const float* Um, const float* Us_packed, const float* g,
                      float* dUm, float* dUs, void* stream);

/* ------------------------------------------------------------------------------------------------
 * conv VAE blocks (model/core/vae.py), NCHW fp32.  Conv geometry: x (B,Ci,H,W), w (Co,Ci,K,K), stride S,
 * padding P, y (B,Co,Ho,Wo).  nn.ConvTranspose2d(Cin_T,Cout_T) with weight (Cin_T,Cout_T,K,K) is the adjoint
 * of the convolution with Co=Cin_T, Ci=Cout_T on the SAME weight buffer:
 *   ConvTranspose2d forward   = gpode_conv2d_bwd_data(gy := its input, bias := its bias)
 *   ConvTranspose2d d/d input = gpode_conv2d_fwd(x := grad_output, bias := NULL)
 *   ConvTranspose2d d/d weight= gpode_conv2d_bwd_weight(x := grad_output, gy := its input)
 * Conv2d (vae.py:53-61) uses them under their own names. */
int gpode_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co,
                     int K, int S, int P, int Ho, int Wo, void* stream);
int gpode_conv2d_bwd_data(const float* gy, const float* w, const float* bias, float* gx, int B, int Ci, int H, int W, int Co,
                          int K, int S, int P, int Ho, int Wo, void* stream);
/* The same with the BatchNorm + ReLU that precedes the ConvTranspose2d folded into its input staging: gy is the RAW output of the
 * previous layer, gy_bn its per-channel table {mean, invstd, gamma, beta} from gpode_bn_stats; the normalised activation is
 * never written to memory (vae.py:113-120: ConvTranspose2d -> BatchNorm2d -> ReLU -> ConvTranspose2d).  Matrix-core
 * specialisations only (decnn.4/7/10); other geometries return an error. */
int gpode_conv2d_bwd_data_bn(const float* gy, const float* gy_bn, const float* w, const float* bias, float* gx, int B, int Ci, int H,
                             int W, int Co, int K, int S, int P, int Ho, int Wo, void* stream);
/* gpode_conv2d_fwd / gpode_conv2d_bwd_weight on a BATCH-STRIDED input: image b starts x_batch_stride floats after image b - 1 and is
 * dense in itself -- the encoder reads frame 0 (or the first frames, order 2) of every sequence of the minibatch X (N,T,1,28,28) in place
 * (odegpvae.py:55-63: `X[:, 0]`), where torch would first copy the slice.  Generic kernels only (input channels not a multiple of 4). */
int gpode_conv2d_fwd_bs(const float* x, size_t x_batch_stride, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co,
                        int K, int S, int P, int Ho, int Wo, void* stream);
int gpode_conv2d_bwd_weight_bs(const float* x, size_t x_batch_stride, const float* gy, float* gw, float* gbias, float* scratch, int B, int Ci,
                               int H, int W, int Co, int K, int S, int P, int Ho, int Wo, void* stream);
/* ConvTranspose2d forward (geometry arguments as gpode_conv2d_bwd_data: the convolution it is the adjoint of; x_bn = NULL or the
 * input's BatchNorm + ReLU table as above) that ALSO produces what the nn.BatchNorm2d in training mode BEHIND it needs (vae.py:107-120:
 * ConvTranspose2d -> BatchNorm2d): batch mean / invstd of the output y, the running-statistics update (momentum; unbiased variance;
 * num_batches_tracked += 1) and the {mean, invstd, gamma, beta} table for the next gpode_conv2d_bwd_data_bn.  The sums are taken while
 * y is stored and combined in a fixed order by the last workgroup to finish: no statistics pass over y, no separate table launch.
 *   gamma, beta (C = output channels): the BatchNorm's affine parameters;  running_mean / running_var / num_batches_tracked may be NULL;
 *   scratch: gpode_convT_fwd_stats_scratch(C) floats;  slot 0..63: launches that may run concurrently need different slots.
 * Matrix-core specialisations only (decnn.1/4/7); other geometries return an error. */
size_t gpode_convT_fwd_stats_scratch(int Cout);
int gpode_convT_fwd_stats(const float* x, const float* x_bn, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co,
                          int K, int S, int P, int Ho, int Wo, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table,
                          float* scratch, int slot, void* stream);
size_t gpode_conv_wgrad_scratch(int B, int Ci, int Co, int K);
int gpode_conv2d_bwd_weight(const float* x, const float* gy, float* gw, float* gbias, float* scratch, int B, int Ci, int H, int W,
                            int Co, int K, int S, int P, int Ho, int Wo, void* stream);
int gpode_conv2d_bwd_weight_bn(const float* x, const float* gy, const float* gy_bn, float* gw, float* gbias, float* scratch, int B, int Ci,
                               int H, int W, int Co, int K, int S, int P, int Ho, int Wo, void* stream);
/* nn.BatchNorm2d in TRAINING mode (batch statistics), optional fused ReLU (vae.py:55-59,113-120).
 * running_* may be NULL (no update); num_batches_tracked (optional, DEVICE int64 scalar, the module buffer) is incremented.
 * gx_chansum (optional, C floats): per-channel sum of gx, i.e. the bias gradient of the
 * convolution that feeds this BatchNorm, produced while gx is written instead of by a separate pass. */
size_t gpode_bn_scratch(int B, int C);
int gpode_bn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                 float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, int B, int C,
                 int HW, int relu, float* scratch, void* stream);
int gpode_bn_bwd(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
                 const float* save_invstd, float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW,
                 int relu, float* scratch, void* stream);
/* Training-mode statistics WITHOUT the output pass: save_mean / save_invstd / running statistics / counter as gpode_bn_fwd, plus
 * table[C][4] = {mean, invstd, gamma, beta} for a consumer that applies the normalisation itself (gpode_conv2d_bwd_*_bn).
 * Backward: gpode_bn_bwd as usual (it needs x, not y). */
int gpode_bn_stats(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd, float* running_mean,
                   float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table, int B, int C, int HW,
                   float* scratch, void* stream);
/* BatchNorm across ranks (data parallelism): the reference's training-mode BatchNorm normalises with the statistics of
 * the WHOLE minibatch (vae.py:55,58,113,116,119).  The layer is split where the ranks exchange <= 2C+1 floats; the library does no
 * communication itself (the host all-gathers with RCCL):
 *   forward   gpode_bn_moments   this shard's per-channel {mean, M2 = sum (x - mean)^2} and its element count, moments[2C+1]
 *             (all-gather moments over the W ranks -> gathered[W][2C+1])
 *             gpode_bn_finalize  rank-ordered combination (Chan et al.), save_mean / save_invstd / running statistics / counter,
 *                                table[C][4] = {mean, invstd, gamma, beta}
 *             gpode_bn_apply     y = relu?(affine(x)) from the table -- or hand the table to gpode_conv2d_bwd_*_bn instead
 *   backward  gpode_bn_bwd_sums  this shard's {sum g, sum g xhat} (g = gy under the ReLU mask), sums[2C]
 *             (all-gather sums -> sums_gathered[W][2C])
 *             gpode_bn_bwd_apply gx with the centring terms sum_r weights[r] sums_gathered[r] / count_all, where weights[r] =
 *                                (rank r's share of the global batch) / (this rank's share) -- the data-parallel gradient average
 *                                weights every rank's loss by its share -- and count_all the global element count; ggamma / gbeta
 *                                stay this shard's sums.  Must be given the scratch gpode_bn_bwd_sums just used. */
int gpode_bn_moments(const float* x, float* moments, int B, int C, int HW, float* scratch, void* stream);
int gpode_bn_finalize(const float* gathered, int nranks, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                      float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, float* table,
                      int C, void* stream);
int gpode_bn_apply(const float* x, const float* table, float* y, int B, int C, int HW, int relu, void* stream);
int gpode_bn_bwd_sums(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
                      const float* save_invstd, float* sums, int B, int C, int HW, int relu, float* scratch, void* stream);
int gpode_bn_bwd_apply(const float* x, const float* gy, const float* gamma, const float* beta, const float* save_mean,
                       const float* save_invstd, const float* sums_gathered, const float* weights, int nranks, float count_all,
                       float* gx, float* ggamma, float* gbeta, float* gx_chansum, int B, int C, int HW, int relu, float* scratch,
                       void* stream);
/* Deferred final reductions.  The backward entry points that end in "sum the partials of nsplit workgroups" (gpode_conv2d_bwd_weight*,
 * gpode_bn_bwd, gpode_bn_bwd_apply, gpode_dec10_bn_bwd_apply, gpode_chan_sum) launch that last step themselves -- unless the call is
 * made between gpode_defer_reductions(1) and gpode_defer_reductions(0): then it is recorded (the caller keeps the scratch buffer alive)
 * and gpode_flush_reductions(stream) runs everything recorded so far in ONE launch; their outputs are valid after that.  This is what
 * autograd's end-of-backward hook of the host mirror does (10-15 launches of a few microseconds of work become one graph node).
 * mode 2 = drop what an aborted backward pass left behind, then as mode 1.  At most 24 pending jobs (further ones launch at once). */
void gpode_defer_reductions(int mode);
int gpode_flush_reductions(void* stream);
/* The decoder's last stage backward, fused: decnn.10 = ConvTranspose2d(16 -> 1, 5, stride 1, padding 2) on 28 x 28 (vae.py:121) fed by
 * ReLU(BatchNorm2d(16)) (vae.py:119-120).  The gradient w.r.t. the normalised activation (25 multiply-adds per element from one
 * 3 KB plane of gy) is recomputed inside both BatchNorm backward passes instead of being written once and read twice:
 *   gpode_dec10_bn_bwd_sums   this shard's {sum g, sum g xhat} (partials stay in scratch; sums[32] too unless NULL)
 *   gpode_dec10_bn_bwd_apply  gc = d loss / d c (c = the BatchNorm's input, B x 16 x 28 x 28), ggamma, gbeta, and the channel sums of gc
 *                             (gc_chansum, may be NULL).  sums_gathered == NULL: statistics of this rank alone; otherwise as
 *                             gpode_bn_bwd_apply.  Must be given the scratch gpode_dec10_bn_bwd_sums just used.
 * c: the BatchNorm input, gy: B x 1 x 28 x 28, w: decnn.10's weight [16][1][5][5]; scratch: gpode_dec10_bn_scratch_floats() floats.
 * Replaces gpode_conv2d_fwd (as the d/d input of decnn.10) + gpode_bn_bwd / gpode_bn_bwd_sums + gpode_bn_bwd_apply for this stage. */
/* gpode_dec10_bn_bwd_sums_wgrad: the sums pass that also produces decnn.10's WEIGHT gradient gw [16][1][5][5] (what
 * gpode_conv2d_bwd_weight_bn computes for this stage in a pass of its own over c) and, unless gbias is NULL, its BIAS gradient gbias[1] =
 * sum gy (gpode_chan_sum): both read c and the gy plane in the same lane layout.  wscratch: gpode_dec10_bn_wgrad_scratch_floats()
 * floats.  The final reductions of gw / gbias obey gpode_defer_reductions like the others. */
int gpode_dec10_bn_wgrad_scratch_floats(void);
int gpode_dec10_bn_bwd_sums_wgrad(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* save_mean,
                                  const float* save_invstd, float* sums, float* gw, float* gbias, int B, float* scratch, float* wscratch,
                                  void* stream);
int gpode_dec10_bn_scratch_floats(void);
int gpode_dec10_bn_bwd_sums(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* save_mean,
                            const float* save_invstd, float* sums, int B, float* scratch, void* stream);
int gpode_dec10_bn_bwd_apply(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, const float* sums_gathered, const float* weights, int nranks, float count_all,
                             float* gc, float* ggamma, float* gbeta, float* gc_chansum, int B, float* scratch, void* stream);
/* nn.BatchNorm2d in EVALUATION mode (running statistics; main.py:157-163 puts the pre-trained VAE in eval()).
 * gy == NULL: out = y = relu?(affine(x)); gy != NULL: out = d/dx (frozen layer: no affine gradients). */
int gpode_bn_eval(const float* x, const float* gy, const float* gamma, const float* beta, const float* running_mean,
                  const float* running_var, float eps, float* out, int B, int C, int HW, int relu, void* stream);
/* out[c] = sum_{b,hw} v[b,c,hw] (bias gradients); scratch: gpode_bn_scratch(B,C) floats. */
int gpode_chan_sum(const float* v, float* out, int B, int C, int HW, float* scratch, void* stream);
/* mode 0: ReLU, 1: sigmoid (vae.py:60,121).  Backward takes the forward OUTPUT y. */
int gpode_act_fwd(const float* x, float* y, size_t n, int mode, void* stream);
int gpode_act_bwd(const float* y, const float* gy, float* gx, size_t n, int mode, void* stream);
/* nn.Linear (vae.py:64,107): x (B,In), w (Out,In).  scratch: gpode_linear_bwd_scratch(B, In, Out) floats, or NULL (the weight
 * gradient of a narrow, tall layer -- the decoder's fc at thousands of rows -- is then summed without row slabs, ~3x slower). */
int gpode_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, void* stream);
size_t gpode_linear_bwd_scratch(int B, int In, int Out);
int gpode_linear_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out,
                     float* scratch, void* stream);
/* nn.Linear on relu(x) with the ReLU folded into the layer (the encoder's Conv2d -> ReLU -> Flatten -> Linear, vae.py:58-61, 72-74): x is
 * the RAW convolution output (B, In); forward y = relu(x) W^T + b, backward gx = (x > 0) (gy W), gw = gy^T relu(x), gb = sum gy.  No
 * activation tensor and no ReLU launches in either direction.  Wide fan-in (In >= 128) only; other shapes return an error. */
int gpode_linear_relu_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, void* stream);
int gpode_linear_relu_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out,
                          void* stream);

/* Decoder.log_prob (vae.py:136-153): ll = log(z) X + log(1-z)(1-X), X broadcast over the leading L copies
 * (nX = numel(X)).  rowsum: the sum([2,3,4,5]) of create_model.py:52-53 fused, rows = L*N, inner = T*C*H*W. */
int gpode_loglik_fwd(const float* X, const float* z, float* ll, size_t n, size_t nX, void* stream);
int gpode_loglik_bwd(const float* X, const float* z, const float* g, float* gz, size_t n, size_t nX, void* stream);
int gpode_loglik_rowsum_fwd(const float* X, const float* z, float* out, size_t rows, size_t inner, size_t nX, void* stream);
int gpode_loglik_rowsum_bwd(const float* X, const float* z, const float* grow, float* gz, size_t rows, size_t inner, size_t nX,
                            void* stream);

/* The last two stages of the training step's forward pass, fused (and their adjoints):
 *   gpode_sigmoid_loglik_fwd  z = sigmoid(a) (vae.py:84, the decoder's nn.Sigmoid) and the Bernoulli log-likelihood of X under z
 *                             (vae.py:136-153) summed per row as create_model.py:49 sums it; every row is cut into nsplit slices,
 *                             part[row][split] (gpode_sigmoid_loglik_splits(rows, inner) proposes nsplit).  X (nX floats) is broadcast
 *                             over the rows * inner / nX copies of it.  Same arithmetic as gpode_act_fwd + gpode_loglik_rowsum_fwd.
 *   gpode_sigmoid_loglik_bwd  ga = grow[row] * d/da, the chain gpode_loglik_rowsum_bwd -> gpode_act_bwd in one pass
 *   gpode_elbo_all_fwd        out[0..3] = {loss, -mean lhood, mean kl, kl_u} (out: 4 + 256 floats, the tail is scratch for the
 *                             sum over a big Us) (create_model.py:61-73) from the nl_values partial sums
 *                             of nl_rows likelihood rows, the encoder's rows hs = (mu | logvar), N x 2q (hv: the velocity half of a
 *                             second-order model, or NULL) -- KL(q(z0) || N(0, I)), create_model.py:47-49 -- and the inducing
 *                             posterior (Um M x Do, Us packed lower triangles Do x M(M+1)/2) -- KL(q(u) || N(0, I)), svpy.py:144-175
 *   gpode_elbo_all_bwd        gradients of the four outputs (device scalars, NULL = 0) -> glrow[nl_rows], ghs / ghv (N x 2q), dUm, dUs */
int gpode_sigmoid_loglik_splits(size_t rows, size_t inner);
int gpode_sigmoid_loglik_fwd(const float* X, const float* a, float* z, float* part, size_t rows, size_t inner, size_t nX, int nsplit,
                             void* stream);
int gpode_sigmoid_loglik_bwd(const float* X, const float* z, const float* grow, float* ga, size_t rows, size_t inner, size_t nX,
                             void* stream);
int gpode_elbo_all_fwd(const float* lpart, int nl_rows, int nl_values, const float* hs, const float* hv, int N, int q, int M, int Do,
                       const float* Um, const float* Us, float nobs, float* out, void* stream);
int gpode_elbo_all_bwd(const float* g_loss, const float* g_nll, const float* g_kl, const float* g_klu, int nl_rows, const float* hs,
                       const float* hv, int N, int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs,
                       float* ghv, float* dUm, float* dUs, void* stream);
/* gpode_elbo_all_bwd and gpode_sigmoid_loglik_bwd in one launch: every likelihood row receives the same gradient, so the gradient of the
 * decoder's logits (ga, n_logits values; X of nX values is broadcast over the Monte-Carlo copies) is produced together with the others. */
int gpode_elbo_all_bwd_ll(const float* g_loss, const float* g_nll, const float* g_kl, const float* g_klu, int nl_rows, const float* hs,
                          const float* hv, int N, int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs,
                          float* ghv, float* dUm, float* dUs, const float* X, const float* z, float* ga, size_t n_logits, size_t nX,
                          void* stream);
/* All device-side noise of a step in one launch (the reference draws on the host with numpy -- kernels.py:13-26,134-137,
 * svpy.py:12-27,94 -- and with torch.randn_like, vae.py:76; `--device_noise` / DeviceNoise draws here instead):
 *   out[0 .. n_normal) ~ N(0,1), out[n_normal .. n_normal + n_uniform) ~ U[0,1): Philox4x32-10 keyed by `seed`, counter = (element
 *   quad, draw number).  state: two 64-bit words in device memory {draw number, 0}; the kernel advances the draw number itself,
 *   so a captured launch yields fresh numbers at every replay and equal (seed, state) give equal numbers on every rank. */
int gpode_noise_fill(float* out, long long n_normal, long long n_uniform, unsigned long long seed, unsigned long long* state, void* stream);
/* ELBO glue on (N,q)-sized tensors (one launch each):
 *   mu / logvar are (N,q) with row stride ld (ld = 2q: the two halves of the encoder's fc output, vae.py:74), gradients
 *   are written with row stride ldg;
 *   reparameterisation z = mu + exp(logvar/2) eps (vae.py:75-78) and its backward;
 *   klrow[n] = sum_d KL(N(mu, exp(logvar/2)) || N(0,1)) (create_model.py:47-49, torch.distributions closed form) and its backward;
 *   out[4] = {loss = -(mean(lhood) nobs - mean(klrow) nobs - kl_u), -mean(lhood), mean(klrow), kl_u} (create_model.py:61-73);
 *   backward: gout[4] (gradients of the four outputs) -> glhood[nl], gklrow[nk], gklu[1]. */
/* The KL(q(z0) || N(0, I)) term riding with the reparameterisation (create_model.py:47-49 next to vae.py:75-78): gpode_reparam_kl_fwd
 * writes z and klpart[ceil(N q / 256)] = partial sums of the KL terms, which gpode_elbo_all_fwd_kl adds up in place of the packed
 * (mu | logvar) rows of gpode_elbo_all_fwd (klv / nkv: the velocity encoder's partials of a second-order model, or NULL / 0);
 * gpode_elbo_all_bwd_ll_kl returns the (uniform) gradient of every partial sum, and gpode_reparam_kl_bwd writes the gradients of
 * (mu, logvar) through z AND through the KL sum in one pass -- no second gradient tensor for autograd to add. */
int gpode_reparam_kl_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, float* klpart, int N, int q, void* stream);
int gpode_reparam_kl_bwd(const float* gz, const float* gklpart, const float* mu, const float* logvar, int ld, const float* eps, float* gmu,
                         float* glogvar, int ldg, int N, int q, void* stream);
int gpode_elbo_all_fwd_kl(const float* lpart, int nl_rows, int nl_values, const float* kls, int nks, const float* klv, int nkv, int N, int M,
                          int Do, const float* Um, const float* Us, float nobs, float* out, void* stream);
int gpode_elbo_all_bwd_ll_kl(const float* g_loss, const float* g_nll, const float* g_kl, const float* g_klu, int nl_rows, int N, int M, int Do,
                             const float* Um, const float* Us, float nobs, float* glrow, float* gkls, int nks, float* gklv, int nkv,
                             float* dUm, float* dUs, const float* X, const float* z, float* ga, size_t n_logits, size_t nX, void* stream);
int gpode_reparam_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, int N, int q, void* stream);
int gpode_reparam_bwd(const float* gz, const float* logvar, int ld, const float* eps, float* gmu, float* glogvar, int ldg, int N, int q,
                      void* stream);
int gpode_normal_kl_fwd(const float* mu, const float* logvar, int ld, float* klrow, int N, int q, void* stream);
int gpode_normal_kl_bwd(const float* grow, const float* mu, const float* logvar, int ld, float* gmu, float* glogvar, int ldg, int N, int q,
                        void* stream);
int gpode_elbo_fwd(const float* lhood, int nl, const float* klrow, int nk, const float* kl_u, float nobs, float* out, void* stream);
int gpode_elbo_bwd(const float* gout, int nl, int nk, float nobs, float* glhood, float* gklrow, float* gklu, void* stream);

/* torch.optim.Adam step (main.py:194,211) over a whole parameter list in one launch.  params/grads/m1/m2:
 * DEVICE arrays of `ntensors` device pointers; offs: DEVICE array of element prefix offsets (offs[0]=0);
 * step: 1-based step count (bias correction).  step_dev (optional, DEVICE int[2] = {updates done so far, 0}): when non-NULL the
 * kernel uses step_dev[0] + 1 instead of `step` and advances step_dev[0] itself (the last workgroup to finish; step_dev[1] is its
 * ticket counter), so that a captured HIP graph of the training step replays with a live count and no separate counter launch. */
int gpode_adam_multi(void* params, void* grads, void* m1, void* m2, const long long* offs, int ntensors, long long total,
                     float lr, float beta1, float beta2, float eps, int step, int* step_dev, void* stream);

/* The data-parallel gradient bucket in one launch: flat[offs[t] + i] = grads[t][i] (same DEVICE tables as gpode_adam_multi).
 * What `loss.backward()` + DDP's bucket copy do in the reference's setting; the bucket is then all-reduced in place (RCCL)
 * and handed to gpode_adam_multi through a pointer table into it. */
int gpode_gather_multi(void* grads, const long long* offs, int ntensors, long long total, float* flat, void* stream);

#ifdef __cplusplus
}
#endif
#endif
