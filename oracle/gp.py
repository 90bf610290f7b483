"""ORACLE (test infrastructure): numpy restatement of the reference GP core, ital/gp.py.

Dense and reference-faithful on purpose: the full N x N kernel matrix is materialised
(reference ital/gp.py:128) and every prediction goes through the stored inverse of
K[T,T] + noise*I, exactly the quantities the reference computes.  The product (ital_amd) uses a
streaming Cholesky-whitened formulation instead; tests compare the two.
"""
import numpy as np


def rbf_kernel(a, b, length_scale, var):
    """v * exp((|a|^2 + |b|^2 - 2 a.b) / (-2 l^2)); squared distances are NOT clamped
    (reference ital/gp.py:410-416)."""
    a = np.atleast_2d(np.asarray(a, dtype=np.float64))
    b = np.atleast_2d(np.asarray(b, dtype=np.float64))
    an = np.sum(a ** 2, axis=-1)
    bn = np.sum(b ** 2, axis=-1)
    return var * np.exp((an[:, None] + bn[None, :] - 2.0 * (a @ b.T)) / (-2.0 * length_scale * length_scale))


def spd_inverse(M):
    """Inverse of a symmetric positive-definite matrix through its Cholesky factor
    (role of invh, reference ital/gp.py:8-37: dpotrf + dpotri + symmetrise)."""
    L = np.linalg.cholesky(M)
    Li = np.linalg.solve(L, np.eye(M.shape[0]))
    inv = Li.T @ Li
    return 0.5 * (inv + inv.T)


class OracleGP:
    """Mirror of reference ital/gp.py:91-436 (fit / update / predict_stored / predict_cov_batch /
    predict / updated_prediction)."""

    def __init__(self, data, length_scale, var=1.0, noise=1e-6):
        self.X = np.array(data, dtype=np.float64)
        self.length_scale = length_scale
        self.var = var
        self.noise = noise
        self.K_all = rbf_kernel(self.X, self.X, length_scale, var)  # gp.py:128
        self.reset()

    def reset(self):  # gp.py:132-138
        self.ind = []
        self.y = self.K = self.K_inv = self.w = None

    def fit(self, ind, y):  # gp.py:141-161
        self.ind = [int(i) for i in ind]
        self.y = np.array(y, dtype=np.float64)
        self.K = self.K_all[np.ix_(self.ind, self.ind)] + self.noise * np.eye(len(self.ind))
        self.K_inv = spd_inverse(self.K)
        self.w = self.K_inv @ self.y
        return self

    def update(self, ind, y):  # gp.py:164-200 (the reference re-inverts from scratch, so does this)
        if len(self.ind) == 0:
            return self.fit(ind, y)
        return self.fit(self.ind + [int(i) for i in ind], np.concatenate((self.y, np.asarray(y, dtype=np.float64))))

    def predict_stored(self, ind=None, cov_mode=None):  # gp.py:203-232
        k_test = self.K_all[self.ind] if ind is None else self.K_all[np.ix_(self.ind, ind)]
        mean = self.w @ k_test
        if cov_mode == "full":
            kk = self.K_all if ind is None else self.K_all[np.ix_(ind, ind)]
            return mean, kk - k_test.T @ (self.K_inv @ k_test)
        if cov_mode == "diag":
            return mean, np.maximum(0, self.var - np.sum(k_test * (self.K_inv @ k_test), axis=0))
        return mean

    def predict_cov_batch(self, base_ind, ind):  # gp.py:235-261
        base_ind = [int(i) for i in base_ind]
        ind = np.asarray(ind, dtype=np.int64)
        t = len(base_ind)
        k_base = self.K_all[np.ix_(self.ind, base_ind)]
        cov_base = self.K_all[np.ix_(base_ind, base_ind)] - k_base.T @ (self.K_inv @ k_base)
        k_test = self.K_all[np.ix_(self.ind, ind)]
        var_test = self.var - np.sum(k_test * (self.K_inv @ k_test), axis=0)  # NOT clamped (gp.py:254)
        cov_bt = self.K_all[np.ix_(base_ind, ind)] - k_base.T @ (self.K_inv @ k_test)
        out = np.empty((len(ind), t + 1, t + 1))
        out[:, :t, :t] = cov_base[None]
        out[:, :t, t] = cov_bt.T
        out[:, t, :t] = cov_bt.T
        out[:, t, t] = var_test
        return out

    def predict(self, X, cov_mode=None):  # gp.py:264-292
        X = np.atleast_2d(np.asarray(X, dtype=np.float64))
        k_test = rbf_kernel(self.X[self.ind], X, self.length_scale, self.var)
        mean = self.w @ k_test
        if cov_mode == "full":
            return mean, rbf_kernel(X, X, self.length_scale, self.var) - k_test.T @ (self.K_inv @ k_test)
        if cov_mode == "diag":
            return mean, np.maximum(0, self.var - np.sum(k_test * (self.K_inv @ k_test), axis=0))
        return mean

    def updated_prediction(self, ind, y, pred_ind, cov_mode=None):
        """Posterior at pred_ind after a simulated update with (ind, y) (gp.py:295-344).  The reference
        extends its stored inverse with a Woodbury step (extend_inv, gp.py:40-87); the matrix it obtains
        is inv(K[T+new, T+new] + noise*I), formed here directly."""
        ext = np.array(self.ind + [int(i) for i in ind], dtype=np.int64)
        pred = np.asarray(pred_ind, dtype=np.int64)
        y_ext = np.concatenate((self.y, np.asarray(y, dtype=np.float64)))
        rows = self.K_all[ext]
        K_ext = rows[:, ext]
        K_ext[np.diag_indices_from(K_ext)] += self.noise
        k_test = rows[:, pred]
        sol = np.linalg.solve(K_ext, np.column_stack((y_ext, k_test)))
        mean = sol[:, 0] @ k_test
        if cov_mode == "full":
            return mean, self.K_all[pred][:, pred] - k_test.T @ sol[:, 1:]
        if cov_mode == "diag":
            return mean, np.maximum(0, self.var - np.sum(k_test * sol[:, 1:], axis=0))
        return mean
