"""Index helpers for the lane-major pack layout (csrc/gp_eval.hpp), used by the tests to read gradients
that the HIP kernels return in pack layout."""
import torch


def cdiv(a, b):
    return (a + b - 1) // b


class PackView:
    def __init__(self, kernel, Di, Do, M, S):
        self.kernel, self.Di, self.Do, self.M, self.S = kernel, Di, Do, M, S
        self.SJ, self.MJ = cdiv(S, 64), cdiv(M, 64)
        if kernel == 'RBF':
            self.RQ, self.RQ2 = cdiv(Di + 2, 4), cdiv(Di + Do, 4)
            self.nuni = Do * Di
        else:
            self.RQ, self.RQ2 = cdiv(2 * Do + 3, 4), cdiv(2 * Do, 4)
            self.nuni = 2 * Do * Do + Do
        self.rff_f4 = self.SJ * Do * self.RQ * 64
        self.ind_f4 = self.MJ * self.RQ2 * 64

    def rff(self, g):
        """-> (S, Do_or_D, 4*RQ): record fields for feature s and record index d (RBF) / i (DF)."""
        t = g[:4 * self.rff_f4].view(self.SJ, self.Do, self.RQ, 64, 4)       # j, d, q, lane, comp
        t = t.permute(0, 3, 1, 2, 4).reshape(self.SJ * 64, self.Do, self.RQ * 4)  # s, d, field
        return t[:self.S]

    def ind(self, g):
        """-> (M, 4*RQ2): record fields for inducing point m."""
        t = g[4 * self.rff_f4:4 * (self.rff_f4 + self.ind_f4)].view(self.MJ, self.RQ2, 64, 4)
        t = t.permute(0, 2, 1, 3).reshape(self.MJ * 64, self.RQ2 * 4)
        return t[:self.M]

    def uni(self, g):
        o = 4 * (self.rff_f4 + self.ind_f4)
        return g[o:o + self.nuni]
