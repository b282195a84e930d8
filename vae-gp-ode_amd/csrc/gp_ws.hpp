// gp_ws.hpp -- layout of the forward cache-build workspace (`ws`), shared with the backward.
#pragma once
#include "gp_eval.hpp"

namespace gp {

static constexpr int NB = 32;

// ---------------------------------------------------------------------------------------------
// workspace layout (floats)
// ---------------------------------------------------------------------------------------------
struct WsLayout {
  size_t info, ell, var, u, u_prior, nu, A, Lmat, Dfac, total;
  int n, np, nblk, batch, nd;
};

// nd = Monte-Carlo draws that share the build (odegpvae.py:41-43): K_uu + jitter I and its factor depend on the parameters only
// (kernels.py:163 / :384), so ONE factorisation serves all of them -- the nd right-hand sides f_prior_l(Z) ride as rows
// n .. n + nd - 1 of the augmented matrix, and u, f_prior(Z), nu are kept per draw.
static inline WsLayout ws_layout(int kernel, int Di, int Do, int M, int S, int nd = 1) {
  WsLayout w;
  w.nd = nd;
  w.n = kernel == 0 ? M : M * Do;
  w.batch = kernel == 0 ? Do : 1;
  w.nblk = cdiv(w.n + nd, NB);
  if (w.nblk >= 32) w.nblk = (w.nblk + 3) / 4 * 4;   // big factors: whole 128-wide panels for the matrix-core kernels (identity padding)
  w.np = w.nblk * NB;
  size_t o = 0;
  auto take = [&](size_t nfl) { size_t at = o; o += (nfl + 3) / 4 * 4; return at; };
  w.info = take(4);
  w.ell = take((size_t)Do * Di);
  w.var = take(Do);
  w.u = take((size_t)nd * M * Do);
  w.u_prior = take((size_t)nd * M * Do);
  w.nu = take((size_t)nd * w.batch * w.n);
  w.A = take((size_t)w.batch * w.np * w.np);
  w.Lmat = take((size_t)w.batch * w.np * w.np);
  w.Dfac = take((size_t)w.batch * w.nblk * NB * NB);
  w.total = o;
  return w;
}


}  // namespace gp
