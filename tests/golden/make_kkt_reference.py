"""An INDEPENDENT restatement of the reference's batch QP, used to certify solver answers.  TEST INFRASTRUCTURE ONLY.

numpy / scipy only.  Imports nothing from oracle/ and nothing from the product package: it takes plain arrays.

What it restates (reference: swarm_planner/include/rbp_planner.hpp):
    build_Q_base :327-347, build_Q_p :349-351, build_Aeq_base :353-405, build_deq :408-432, build_dlq :435-511 (box and
    RSFC-normal selection by end time), build_dummy :513-549, setBatch / isQuadInBatch :849-881 and populatebyrow :551-688
    -- in the REFERENCE's variable order  x[k * offset_dim + bi * offset_quad + m * (n+1) + i]  (:552-561), as explicit
    matrices  (Q, Aeq, deq, G, h):   minimise x'Qx (no 1/2, :582-605)   s.t.  Aeq x = deq,  G x <= h.

What it certifies.  The batch QP is convex with a unique optimum on its feasible set, so an answer x is THE
answer iff the KKT conditions hold.  `certify_batch` takes a solver's x and
    1. checks primal feasibility of x on every row;
    2. reads the active set  A = {rows with slack <= tau}  off x  (the only thing taken from the solver);
    3. RE-DERIVES the optimum of  min x'Qx  s.t. Aeq x = deq, G_A x = h_A  by dense float64 linear algebra (null-space
       method with SVD rank decisions: a path that shares nothing with the interior-point / block-Cholesky / Lawson-Hanson
       code of the GPU kernel or of oracle/planner.c), checks that this point is feasible for ALL rows and
    4. that multipliers lambda >= 0 on A exist (non-negative least squares on the reduced stationarity equations).
    Steps 3+4 prove x_as is the optimum; ||x - x_as||_inf is then the solver's true forward error.
`certify_plan` walks the Gauss-Seidel schedule of solveQP (:140-203) for plan/iteration = 1: batch l is solved against the
final control points of batches < l and the initial `dummy` of batches > l, which is reconstructible from the final answer.

Run as a script it writes tests/golden/kkt_reference_c1.npz: the explicit matrices of the 4-agent C1 joint QP, so that the
restatement itself is pinned against silent edits (tests/test_kkt_reference.py compares).
"""
import numpy as np
from scipy.optimize import nnls

N_DEG, PHI, OUTDIM = 5, 3, 3


def Q_base():
    """rbp_planner.hpp:330-335"""
    return np.array([[720, -1800, 1200, 0, 0, -120],
                     [-1800, 4800, -3600, 0, 600, 0],
                     [1200, -3600, 3600, -1200, 0, 0],
                     [0, 0, -1200, 3600, -3600, 1200],
                     [0, 600, 0, -3600, 4800, -1800],
                     [-120, 0, 0, 1200, -1800, 720]], dtype=np.float64)


def A_0_T():
    """rbp_planner.hpp:362-374"""
    A0 = np.array([[1, 0, 0, 0, 0, 0], [-1, 1, 0, 0, 0, 0], [1, -2, 1, 0, 0, 0], [-1, 3, -3, 1, 0, 0], [1, -4, 6, -4, 1, 0],
                   [-1, 5, -10, 10, -5, 1]], dtype=np.float64)
    AT = np.array([[0, 0, 0, 0, 0, 1], [0, 0, 0, 0, -1, 1], [0, 0, 0, 1, -2, 1], [0, 0, -1, 3, -3, 1], [0, 1, -4, 6, -4, 1],
                   [-1, 5, -10, 10, -5, 1]], dtype=np.float64)
    return A0, AT


def Aeq_base(T):
    """rbp_planner.hpp:353-405: (2 phi + (M-1) phi) x M (n+1)"""
    T = np.asarray(T, np.float64)
    M, n, phi = len(T) - 1, N_DEG, PHI
    A0, AT = A_0_T()
    A = np.zeros((2 * phi + (M - 1) * phi, M * (n + 1)))
    nn = 1
    for i in range(phi):  # :380-387
        A[i, 0:n + 1] = (T[1] - T[0]) ** (-i) * nn * A0[i]
        A[phi + i, (n + 1) * (M - 1):(n + 1) * M] = (T[M] - T[M - 1]) ** (-i) * nn * AT[i]
        nn *= (n - i)
    for m in range(1, M):  # :390-399
        nn = 1
        for j in range(phi):
            r = 2 * phi + phi * (m - 1) + j
            A[r, (n + 1) * (m - 1):(n + 1) * m] = (T[m] - T[m - 1]) ** (-j) * nn * AT[j]
            A[r, (n + 1) * m:(n + 1) * (m + 1)] = -(T[m + 1] - T[m]) ** (-j) * nn * A0[j]
            nn *= (n - j)
    return A


def deq_agent(start9, goal9, M):
    """rbp_planner.hpp:408-432 for one agent: (2 phi + (M-1) phi) x 3"""
    d = np.zeros((2 * PHI + (M - 1) * PHI, OUTDIM))
    for k in range(OUTDIM):
        d[0, k], d[1, k], d[2, k] = start9[k], start9[k + 3], start9[k + 6]
        d[PHI, k], d[PHI + 1, k], d[PHI + 2, k] = goal9[k], goal9[k + 3], goal9[k + 6]
    return d


def select_boxes(T, sfc_box, sfc_time, sfc_count):
    """rbp_planner.hpp:447-469: per (agent, segment) the first box whose end time is not before T[m+1].
    Returns lo, hi [N][M][3]."""
    N, M = len(sfc_count), len(T) - 1
    lo, hi = np.zeros((N, M, 3)), np.zeros((N, M, 3))
    for qi in range(N):
        bi = 0
        for m in range(M):
            while bi < sfc_count[qi] and sfc_time[qi, bi] < T[m + 1]:
                bi += 1
            b = min(bi, sfc_count[qi] - 1)  # the reference would read past the end here; guarded like every restatement
            lo[qi, m], hi[qi, m] = sfc_box[qi, b, 0:3], sfc_box[qi, b, 3:6]
    return lo, hi


def select_normals(T, rsfc_normal, rsfc_time):
    """rbp_planner.hpp:483-497: per (pair, segment) the first RSFC entry whose time is not before T[m+1].  With
    RSFC times = T[1..M] (rbp_corridor.hpp:390) this is the identity; kept literal."""
    npair, M = rsfc_normal.shape[0], len(T) - 1
    out = np.zeros((npair, M, 3))
    for m in range(M):
        ri = 0
        while ri < M and rsfc_time[ri] < T[m + 1]:
            ri += 1
        out[:, m] = rsfc_normal[:, min(ri, M - 1)].astype(np.float64)
    return out


def build_dummy(init_traj):
    """rbp_planner.hpp:513-549: [N][6M][3]; first three control points of segment m = waypoint m, last three = waypoint m+1"""
    tr = np.asarray(init_traj, np.float32).astype(np.float64)
    N, P, _ = tr.shape
    M = P - 1
    d = np.zeros((N, 6 * M, 3))
    for m in range(M):
        d[:, 6 * m:6 * m + 3] = tr[:, m:m + 1]
        d[:, 6 * m + 3:6 * m + 6] = tr[:, m + 1:m + 2]
    return d


def batches(N, sequential, batch_size, batch_iter):
    """setBatch :849-872.  Returns (list of agent lists, number of batches solved per pass)."""
    if sequential:
        bmax = int(np.ceil(N / batch_size))
        if batch_iter < 0 or batch_iter > bmax:
            batch_iter = bmax
    else:
        bmax = int(np.ceil(N / batch_size))  # sized with the ORIGINAL batch_size (:850), then batch_size := N
        batch_size, batch_iter = N, 1
    b = [[] for _ in range(max(bmax, 1))]
    for qi in range(N):
        b[qi // batch_size].append(qi)
    return b, batch_iter


def pair_index(N, qi, qj):
    return qi * N - qi * (qi + 1) // 2 + (qj - qi - 1)


class BatchQP:
    """populatebyrow :551-688 for batch l as explicit dense/structured data in the reference's variable order."""

    def __init__(self, T, start, goal, radius, lo, hi, normals, dummy, batch):
        T = np.asarray(T, np.float64)
        N, M = start.shape[0], len(T) - 1
        n1, oq = N_DEG + 1, (N_DEG + 1) * M
        nb = len(batch)
        self.batch, self.M, self.nb = list(batch), M, nb
        offset_quad, offset_dim = oq, nb * oq
        self.nx = OUTDIM * offset_dim
        var = lambda k, bi, j: k * offset_dim + bi * offset_quad + j
        self.var = var
        # objective: block diagonal Q_p per (k, bi, m)   :582-605
        Qb = Q_base()
        self.Qseg = np.stack([Qb * (T[m + 1] - T[m]) ** (-2 * PHI + 1) for m in range(M)])  # [M][6][6]
        # equalities :608-622 -- Aeq_base per (k, bi), rhs from deq
        self.Ab = Aeq_base(T)
        ne = self.Ab.shape[0]
        self.deq = np.zeros(OUTDIM * nb * ne)
        for k in range(OUTDIM):
            for bi, qi in enumerate(batch):
                self.deq[(k * nb + bi) * ne:(k * nb + bi + 1) * ne] = deq_agent(start[qi], goal[qi], M)[:, k]
        # inequalities, G x <= h, as (row -> list of (col, val)) in COO form
        rows, cols, vals, h = [], [], [], []
        r = 0
        for k in range(OUTDIM):  # SFC :626-635
            for bi, qi in enumerate(batch):
                for j in range(oq):
                    m = j // n1
                    rows += [r, r + 1],
                    cols += [var(k, bi, j), var(k, bi, j)],
                    vals += [1.0, -1.0],
                    h += [hi[qi, m, k], -lo[qi, m, k]]
                    r += 2
        rows = [x for pr in rows for x in pr]
        cols = [x for pr in cols for x in pr]
        vals = [x for pr in vals for x in pr]
        self.n_sfc = r
        inb = {qi: bi for bi, qi in enumerate(batch)}
        for qi in range(N):  # RSFC :638-684
            for qj in range(qi + 1, N):
                bi, bj = inb.get(qi, -1), inb.get(qj, -1)
                if bi < 0 and bj < 0:
                    continue
                nv = normals[pair_index(N, qi, qj)]  # [M][3]
                rr = radius[qi] + radius[qj]
                for j in range(oq):
                    nrm = nv[j // n1]
                    if bi >= 0 and bj < 0:    # n.(dummy_j - x_i) >= rr   ->   n.x_i <= n.dummy_j - rr
                        for k in range(OUTDIM):
                            rows.append(r), cols.append(var(k, bi, j)), vals.append(nrm[k])
                        h.append(float(nrm @ dummy[qj, j]) - rr)
                    elif bi < 0 and bj >= 0:  # n.(x_j - dummy_i) >= rr   ->  -n.x_j <= -n.dummy_i - rr
                        for k in range(OUTDIM):
                            rows.append(r), cols.append(var(k, bj, j)), vals.append(-nrm[k])
                        h.append(-float(nrm @ dummy[qi, j]) - rr)
                    else:                     # n.(x_j - x_i) >= rr       ->   n.x_i - n.x_j <= -rr
                        for k in range(OUTDIM):
                            rows.append(r), cols.append(var(k, bi, j)), vals.append(nrm[k])
                            rows.append(r), cols.append(var(k, bj, j)), vals.append(-nrm[k])
                        h.append(-rr)
                    r += 1
        self.n_ineq = r
        self.g_rows, self.g_cols, self.g_vals = np.array(rows), np.array(cols), np.array(vals, np.float64)
        self.h = np.array(h, np.float64)
        self.count_x, self.count_eq, self.count_lq = self.nx, OUTDIM * nb * ne, self.n_ineq  # :579, :623, :687

    # ---- operators -----------------------------------------------------------------------------------------------
    def x_of(self, ctrl):
        """ctrl [N][3][6M] (the C ABI's layout) -> x in the reference's variable order"""
        return np.concatenate([ctrl[qi, k] for k in range(OUTDIM) for qi in self.batch])

    def Qx(self, x):
        X = x.reshape(OUTDIM * self.nb, self.M, 6)
        return np.einsum("mij,umj->umi", self.Qseg, X).reshape(-1)

    def objective(self, x):
        return float(x @ self.Qx(x))

    def G_dot(self, x):
        out = np.zeros(self.n_ineq)
        np.add.at(out, self.g_rows, self.g_vals * x[self.g_cols])
        return out

    def G_dense_rows(self, idx):
        """dense rows idx of G"""
        pos = {r: i for i, r in enumerate(idx)}
        sel = np.isin(self.g_rows, idx)
        Gd = np.zeros((len(idx), self.nx))
        for r, c, v in zip(self.g_rows[sel], self.g_cols[sel], self.g_vals[sel]):
            Gd[pos[r], c] += v
        return Gd

    def eq_residual(self, x):
        ne = self.Ab.shape[0]
        X = x.reshape(OUTDIM * self.nb, -1)
        return (X @ self.Ab.T).reshape(-1) - self.deq


def _nullspace(A, rtol=1e-11):
    U, s, Vt = np.linalg.svd(A, full_matrices=True)
    rank = int((s > rtol * s[0]).sum()) if s.size else 0
    return Vt[rank:].T, rank


def certify_batch(qp: BatchQP, x, tau=2e-8):
    """see the module docstring.  Returns a dict of residuals; nothing is asserted here."""
    nb, M = qp.nb, qp.M
    nu = OUTDIM * nb
    # equality elimination: x = x0 + Z u per (k, bi) block with the SAME Aeq_base
    Zb, _ = _nullspace(qp.Ab)                       # 6M x (6M - 3(M+1))
    Ab_pinv = np.linalg.pinv(qp.Ab)
    ne, oq, nz = qp.Ab.shape[0], qp.Ab.shape[1], Zb.shape[1]
    x0 = (qp.deq.reshape(nu, ne) @ Ab_pinv.T).reshape(-1)
    slack = qp.h - qp.G_dot(x)
    out = {"viol_eq": float(np.abs(qp.eq_residual(x)).max()), "viol_ineq": float(max(0.0, -slack.min())),
           "objective": qp.objective(x)}
    act = np.nonzero(slack <= tau)[0]
    out["n_active"] = int(len(act))
    # reduced problem in u:  min (x0 + Z u)' Q (x0 + Z u)  s.t.  (G_A Z) u = h_A - G_A x0
    def Zt(v):   # Z' v
        return (v.reshape(nu, oq) @ Zb).reshape(-1)
    def Zm(u):   # Z u
        return (u.reshape(nu, nz) @ Zb.T).reshape(-1)
    # reduced Hessian: block diagonal, the same block for every (k, bi):  Zb' (2 Q) Zb
    Qfull = np.zeros((oq, oq))
    for m in range(M):
        Qfull[6 * m:6 * m + 6, 6 * m:6 * m + 6] = qp.Qseg[m]
    Hb = Zb.T @ (2 * Qfull) @ Zb
    g0 = Zt(2 * qp.Qx(x0))
    if len(act):
        GA = qp.G_dense_rows(act)
        GAZ = (GA.reshape(len(act), nu, oq) @ Zb).reshape(len(act), nu * nz)
        rhs = qp.h[act] - GA @ x0
        u_p, *_ = np.linalg.lstsq(GAZ, rhs, rcond=1e-12)
        out["active_rows_inconsistency"] = float(np.abs(GAZ @ u_p - rhs).max())
        Nn, rank = _nullspace(GAZ)
        out["active_rank"] = rank
    else:
        GAZ = np.zeros((0, nu * nz))
        u_p = np.zeros(nu * nz)
        Nn = np.eye(nu * nz)
    Hu = lambda u: (u.reshape(nu, nz) @ Hb).reshape(-1)   # H u (H block diagonal, Hb symmetric)
    HN = np.stack([Hu(Nn[:, i]) for i in range(Nn.shape[1])], axis=1) if Nn.shape[1] else np.zeros((nu * nz, 0))
    NHN = Nn.T @ HN
    gred = Nn.T @ (Hu(u_p) + g0)
    w = np.linalg.solve(NHN, -gred) if Nn.shape[1] else np.zeros(0)
    u_as = u_p + Nn @ w
    x_as = x0 + Zm(u_as)
    out["x_as"] = x_as
    out["forward_error"] = float(np.abs(x - x_as).max())
    slack_as = qp.h - qp.G_dot(x_as)
    out["x_as_viol_ineq"] = float(max(0.0, -slack_as.min()))
    out["x_as_viol_eq"] = float(np.abs(qp.eq_residual(x_as)).max())
    out["objective_as"] = qp.objective(x_as)
    # multipliers: (G_A Z)' lambda = -(H u_as + g0), lambda >= 0
    grad = Hu(u_as) + g0
    gscale = float(np.abs(Zt(2 * np.abs(qp.Qx(np.abs(x_as))))).max()) + 1.0
    if len(act):
        lam, rn = nnls(GAZ.T, -grad, maxiter=20 * GAZ.shape[0] + 200)
        out["stationarity"] = float(np.abs(GAZ.T @ lam + grad).max() / gscale)
        out["lambda_max"] = float(lam.max()) if lam.size else 0.0
        out["complementarity"] = float((lam * np.maximum(slack_as[act], 0)).max()) if lam.size else 0.0
    else:
        out["stationarity"] = float(np.abs(grad).max() / gscale)
        out["lambda_max"] = 0.0
        out["complementarity"] = 0.0
    return out


def certify_plan(T0, init_traj, start, goal, radius, sfc_box, sfc_time, sfc_count, rsfc_normal, rsfc_time, ctrl, sequential,
                 batch_size, batch_iter, tau=2e-8, only_batches=None, ctrl_before_pass=None):
    """every batch QP of solveQP's LAST pass, reconstructed from the FINAL control points `ctrl` ([N][3][6M]): frozen agents of
    batches < l at their final values, of batches > l at the values they had when the pass began -- build_dummy for plan/iteration
    = 1 (:116), the control points after the previous pass for plan/iteration > 1 (`ctrl_before_pass`, [N][3][6M]: `dummy` is only
    ever overwritten by :183-185).  Returns one report per batch."""
    T0 = np.asarray(T0, np.float64)
    N = start.shape[0]
    lo, hi = select_boxes(T0, sfc_box, sfc_time, sfc_count)
    normals = select_normals(T0, rsfc_normal, rsfc_time)
    dummy = build_dummy(init_traj)
    if ctrl_before_pass is not None:
        for qi in range(N):
            dummy[qi] = np.asarray(ctrl_before_pass)[qi].T
    bl, biter = batches(N, sequential, batch_size, batch_iter)
    reports = []
    for l in range(biter):
        for qi in bl[l]:  # never read for in-batch agents; set only so that the state after the loop equals solveQP's :183-185
            pass
        if only_batches is None or l in only_batches:
            qp = BatchQP(T0, start, goal, radius, lo, hi, normals, dummy, bl[l])
            rep = certify_batch(qp, qp.x_of(ctrl), tau)
            rep["batch"] = l
            rep["count_x"], rep["count_eq"], rep["count_lq"] = qp.count_x, qp.count_eq, qp.count_lq
            reports.append(rep)
        for qi in bl[l]:  # dummy <- vals  (:183-185)
            dummy[qi] = ctrl[qi].T
    return reports


if __name__ == "__main__":
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "c1_4agents_empty_joint.npz"))
    import json
    # mission geometry of the C1 case: start/goal come with the golden's mission file; only arrays are used here
    raise SystemExit("the C1 matrix fixture is written by tests/golden/make_golden.py --kkt (needs the mission loader)")


# ---------------------------------------------------------------------------------------------------------------------------------
# timeScale (rbp_planner.hpp:209-266 with :708-847), restated with BOTH root rules, to measure the one conscious deviation of the
# product and the oracle: roots_derivative (:727-754) inspects the first `i` eigenvalues of Eigen's companion-matrix
# solver (`for j < i`, :747; i = 2 for the velocity test, whose companion matrix has order 3), the product uses all real roots.
# Eigen is absent here; LAPACK's eigenvalue order (numpy.linalg.eigvals) stands in for Eigen's.
# ---------------------------------------------------------------------------------------------------------------------------------
def _coef_der(c):
    """c[0..5]: descending powers of one segment (rbp_planner.hpp:708-719) -> coef_der[i][q], i = 0..3"""
    n = 5
    out = np.zeros((4, n + 1))
    for i in range(4):
        for j in range(n + 1):
            if i <= j:
                f = 1
                for t in range(i):
                    f *= (j - t)
                out[i, n - j] = f * c[n - j]
    return out


def _roots_reference_rule(cd, i):
    """roots_derivative(i, coef_der) as written (:727-754): only the first i eigenvalues are looked at"""
    n = 5
    n_der = n - i
    while n_der > 0 and cd[i, n - i - n_der] == 0:
        n_der -= 1
    if n_der == 0:
        return []
    A = np.zeros((n_der, n_der))
    for j in range(n_der):
        if j < n_der - 1:
            A[j + 1, j] = 1
        A[0, j] = -cd[i, n - i - n_der + j + 1] / cd[i, n - i - n_der]
    ev = np.linalg.eigvals(A)
    return [float(np.real(ev[j])) for j in range(min(i, len(ev))) if np.imag(ev[j]) == 0]



def eigen33_eigenvalues(A):
    """Eigenvalues of a small real upper-Hessenberg matrix in the order Eigen 3.3.x's EigenSolver returns them, restated with whole-matrix
    reflections from the published algorithm (RealSchur.h: scale by max|a_ij|, Francis double-shift QR, deflation test
    |t(k,k-1)| <= eps (|t(k-1,k-1)| + |t(k,k)|), exceptional shifts at local iterations 10 and 30, a trailing 2 x 2 block with real
    eigenvalues is rotated to triangular form; EigenSolver.h: read T from the top).  Independent of oracle/planner.c's element-wise
    restatement of the same algorithm -- the two are compared in tests/test_timescale_rule.py."""
    A = np.array(A, dtype=np.float64)
    n = A.shape[0]
    scale = np.abs(A).max()
    if scale < np.finfo(float).tiny:
        return [complex(0.0)] * n
    T = A / scale
    eps = np.finfo(float).eps

    def reflector(v):
        """(essential part, tau, beta) of the Householder reflection H = I - tau [1; ess][1; ess]' with H v = beta e1 (Householder.h)"""
        tail = float(np.dot(v[1:], v[1:]))
        if tail <= np.finfo(float).tiny:
            return np.zeros(len(v) - 1), 0.0, float(v[0])
        beta = np.sqrt(v[0] * v[0] + tail)
        if v[0] >= 0:
            beta = -beta
        return v[1:] / (v[0] - beta), (beta - v[0]) / beta, beta

    def H_of(ess, tau, k):
        u = np.zeros(n)
        u[k] = 1.0
        u[k + 1:k + 1 + len(ess)] = ess
        return np.eye(n) - tau * np.outer(u, u)

    iu, it, total, exshift = n - 1, 0, 0, 0.0
    while iu >= 0:
        il = iu
        while il > 0 and not (abs(T[il, il - 1]) <= eps * (abs(T[il - 1, il - 1]) + abs(T[il, il]))):
            il -= 1
        if il == iu:
            T[iu, iu] += exshift
            if iu > 0:
                T[iu, iu - 1] = 0.0
            iu, it = iu - 1, 0
        elif il == iu - 1:
            p = 0.5 * (T[iu - 1, iu - 1] - T[iu, iu])
            q = p * p + T[iu, iu - 1] * T[iu - 1, iu]
            T[iu, iu] += exshift
            T[iu - 1, iu - 1] += exshift
            if q >= 0:
                z = np.sqrt(abs(q))
                gp, gq = (p + z if p >= 0 else p - z), T[iu, iu - 1]
                if gq == 0:
                    c, sn = (-1.0 if gp < 0 else 1.0), 0.0
                elif gp == 0:
                    c, sn = 0.0, (1.0 if gq < 0 else -1.0)
                elif abs(gp) > abs(gq):
                    t = gq / gp
                    u = np.sqrt(1 + t * t) * (-1.0 if gp < 0 else 1.0)
                    c = 1 / u
                    sn = -t * c
                else:
                    t = gp / gq
                    u = np.sqrt(1 + t * t) * (-1.0 if gq < 0 else 1.0)
                    sn = -1 / u
                    c = -t * sn
                G = np.eye(n)
                G[iu - 1, iu - 1], G[iu - 1, iu], G[iu, iu - 1], G[iu, iu] = c, -sn, sn, c   # rows: x' = c x - s y, y' = s x + c y
                T[:, iu - 1:] = (G @ T)[:, iu - 1:]
                T[:iu + 1, :] = (T @ G.T)[:iu + 1, :]
                T[iu, iu - 1] = 0.0
            if iu > 1:
                T[iu - 1, iu - 2] = 0.0
            iu, it = iu - 2, 0
        else:
            sh = [T[iu, iu], T[iu - 1, iu - 1], T[iu, iu - 1] * T[iu - 1, iu]]
            if it == 10:
                exshift += sh[0]
                T[np.arange(iu + 1), np.arange(iu + 1)] -= sh[0]
                sd = abs(T[iu, iu - 1]) + abs(T[iu - 1, iu - 2])
                sh = [0.75 * sd, 0.75 * sd, -0.4375 * sd * sd]
            if it == 30:
                sd = (sh[1] - sh[0]) / 2.0
                sd = sd * sd + sh[2]
                if sd > 0:
                    sd = np.sqrt(sd)
                    if sh[1] < sh[0]:
                        sd = -sd
                    sd = sd + (sh[1] - sh[0]) / 2.0
                    sd = sh[0] - sh[2] / sd
                    exshift += sd
                    T[np.arange(iu + 1), np.arange(iu + 1)] -= sd
                    sh = [0.964, 0.964, 0.964]
            it, total = it + 1, total + 1
            if total > 40 * n:
                raise RuntimeError("no convergence")
            im = iu - 2
            while True:
                Tmm = T[im, im]
                r, sd = sh[0] - Tmm, sh[1] - Tmm
                v = np.array([(r * sd - sh[2]) / T[im + 1, im] + T[im, im + 1], T[im + 1, im + 1] - Tmm - r - sd, T[im + 2, im + 1]])
                if im == il:
                    break
                lhs = T[im, im - 1] * (abs(v[1]) + abs(v[2]))
                rhs = v[0] * (abs(T[im - 1, im - 1]) + abs(Tmm) + abs(T[im + 1, im + 1]))
                if abs(lhs) < eps * rhs:
                    break
                im -= 1
            for k in range(im, iu - 1):
                first = k == im
                w = v if first else T[k:k + 3, k - 1].copy()
                ess, tau, beta = reflector(w)
                if beta != 0:
                    if first and k > il:
                        T[k, k - 1] = -T[k, k - 1]
                    elif not first:
                        T[k, k - 1] = beta
                    H = H_of(ess, tau, k)
                    T[k:k + 3, k:] = (H @ T)[k:k + 3, k:]
                    rows = min(iu, k + 3) + 1
                    T[:rows, k:k + 3] = (T @ H)[:rows, k:k + 3]
            ess, tau, beta = reflector(T[iu - 1:iu + 1, iu - 2].copy())
            if beta != 0:
                T[iu - 1, iu - 2] = beta
                H = H_of(ess, tau, iu - 1)
                T[iu - 1:iu + 1, iu - 1:] = (H @ T)[iu - 1:iu + 1, iu - 1:]
                T[:iu + 1, iu - 1:iu + 1] = (T @ H)[:iu + 1, iu - 1:iu + 1]
            for i in range(im + 2, iu + 1):
                T[i, i - 2] = 0.0
                if i > im + 2:
                    T[i, i - 3] = 0.0
    T = T * scale
    ev, i = [], 0
    while i < n:
        if i == n - 1 or T[i + 1, i] == 0:
            ev.append(complex(T[i, i], 0.0))
            i += 1
        else:
            p = 0.5 * (T[i, i] - T[i + 1, i + 1])
            t0, t1 = T[i + 1, i], T[i, i + 1]
            mx = max(abs(p), abs(t0), abs(t1))
            z = mx * np.sqrt(abs((p / mx) * (p / mx) + (t0 / mx) * (t1 / mx)))
            ev += [complex(T[i + 1, i + 1] + p, z), complex(T[i + 1, i + 1] + p, -z)]
            i += 2
    return ev


def companion(c):
    """the reference's companion matrix of c[0] t^d + ... + c[d] (:737-744)"""
    d = len(c) - 1
    A = np.zeros((d, d))
    for j in range(d):
        if j < d - 1:
            A[j + 1, j] = 1
        A[0, j] = -c[j + 1] / c[0]
    return A


def _roots_eigen33_rule(cd, i):
    """roots_derivative(i, coef_der) as written (:727-754) with Eigen 3.3's eigenvalue order (eigen33_eigenvalues)"""
    n = 5
    p = np.trim_zeros(cd[i, :n - i + 1], "f")
    if len(p) <= 1:
        return []
    ev = eigen33_eigenvalues(companion(p))
    return [ev[j].real for j in range(min(i, len(ev))) if ev[j].imag == 0]


def _roots_all_real(cd, i):
    n = 5
    p = cd[i, :n - i + 1]
    p = np.trim_zeros(p, "f")
    if len(p) <= 1:
        return []
    return [float(np.real(r)) for r in np.roots(p) if abs(np.imag(r)) < 1e-12 * max(1.0, abs(r))]


def time_scale_of(coef, T, max_vel, max_acc, rule):
    """timeScale's factor (:209-233) for coef [N][3][6M] (descending powers per segment); rule: 'reference' (first two eigenvalues,
    LAPACK's order) | 'eigen33' (first two eigenvalues, Eigen 3.3's order: rbp_param.timescale_rule = 1) | 'all_real' (= 0)"""
    N, _, oq = coef.shape
    M = oq // 6
    roots = {"reference": _roots_reference_rule, "eigen33": _roots_eigen33_rule, "all_real": _roots_all_real}[rule]
    ts_all = 1.0
    for qi in range(N):
        for k in range(3):
            for m in range(M):
                cd = _coef_der(coef[qi, k, 6 * m:6 * m + 6])
                dt = T[m + 1] - T[m]
                cand = roots(cd, 2) + [0.0, dt]
                vel_max, t_max = 0.0, 0.0
                for t in cand:
                    if t < 0 or t > dt:
                        continue
                    v = abs(sum(cd[1, i] * t ** (4 - i) for i in range(5)))
                    if vel_max < v:
                        vel_max, t_max = v, t
                sc = 1.0
                while vel_max > max_vel[qi, k]:
                    sc *= 1.1
                    vel_max = abs(sum(cd[1, i] * (1 / sc) ** (5 - i) * t_max ** (4 - i) for i in range(5)))
                ts_all = max(ts_all, sc)
                a, b, c = cd[3, 0], cd[3, 1], cd[3, 2]
                D = b * b - 4 * a * c
                cand = [0.0, dt]
                if D >= 0 and a != 0:
                    cand += [(-b + np.sqrt(D)) / (2 * a), (-b - np.sqrt(D)) / (2 * a)]
                elif a == 0 and b != 0:
                    cand.append(-c / b)
                acc_max, t_max = 0.0, 0.0
                for t in cand:
                    if t < 0 or t > dt:
                        continue
                    v = abs(sum(cd[2, i] * t ** (3 - i) for i in range(4)))
                    if acc_max < v:
                        acc_max, t_max = v, t
                sc = 1.0
                while acc_max > max_acc[qi, k]:
                    sc *= 1.1
                    acc_max = abs(sum(cd[2, i] * (1 / sc) ** (5 - i) * t_max ** (3 - i) for i in range(4)))
                ts_all = max(ts_all, sc)
    return ts_all
