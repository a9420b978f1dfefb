"""ORACLE — TEST INFRASTRUCTURE ONLY (see lo_math.h).  CPU restatement of the reference back-end's sliding-window problem
(BASELINE configs[4]): the pieces of `optimizeSlidingWindowWithLandMark` that are NOT lidar residuals, plus a dense
Levenberg-Marquardt loop with Ceres 2.0's default rules, so that a 3-keyframe window (t, q, speed-bias per keyframe = 45 local
dimensions) can be solved on the host with the lidar blocks coming either from the oracle's per-residual rows or from the GPU's Gram
records (`lili_s2m_linearize` + `lili_gram_to_factor`, the binding of include/lili_ceres_adapter.h).

Restated from (L/ = /root/reference/LiLi-OM/):
  Preintegration            L/include/factors/Preintegration.h:23-211  (mid-point integration, F / V covariance propagation, evaluate)
  imu_factor                L/include/factors/ImuFactor.h:18-144       (residual, sqrt-information from LLT(cov^-1), six Jacobian blocks)
  speed_bias_prior          L/include/factors/PriorFactor.h:13-23
  Marginalization           L/src/MarginalizationFactor.cpp:128-202 (Schur complement through two eigen-decompositions),
                            :228-287 (MarginalizationFactor::Evaluate)
  problem assembly          L/src/BackendFusion.cpp:843-992 (parameter blocks t[3], q[4] wxyz + QuaternionParameterization, speed-bias[9];
                            marginalisation prior, speed-bias priors, IMU factors between consecutive keyframes, lidar blocks with CauchyLoss(1))
  ceres_lm                  Ceres Solver 2.0 defaults as the reference leaves them (third-party, restated from the published
                            algorithm: TRUST_REGION / LEVENBERG_MARQUARDT, Jacobi scaling, radius update 1 / max(1/3, 1 - (2 rho - 1)^3), ...)

PARITY: Preintegration and imu_factor are pinned against the reference's own headers compiled unmodified (oracle/_ref/libref_imu.so,
tests/test_window_cpu.py: state, covariance, residual and all six Jacobian blocks to 1e-9 relative — the two sides use different
(both valid) summation orders for the 15x15 products, so not bit for bit).  The marginalisation and the LM loop are third-party-shaped
dense algebra (Eigen's SelfAdjointEigenSolver on run-time sizes, Ceres' minimiser): checked against their defining properties
(Schur complement = the marginal of the quadratic; LM reaches the least-squares minimum with Ceres' acceptance rule), parity unpinned.
"""
import numpy as np

O_P, O_R, O_V, O_BA, O_BG = 0, 3, 6, 9, 12


# ---------------------------------------------------------------- quaternions (w, x, y, z), Eigen 3.3 formulas
def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3], a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def qinv(q):
    n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0]
    return np.array([q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2])


def qnormalized(q):
    return q / np.sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3] + q[0] * q[0])


def qrot(q, v):
    u = q[1:]
    uv = np.cross(u, v)
    uv = uv + uv
    return v + q[0] * uv + np.cross(u, uv)


def qmat(q):          # toRotationMatrix
    w, x, y, z = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def delta_q(theta):   # utils/math_tools.h:125-138 — un-normalised small-angle quaternion
    h = np.asarray(theta, np.float64) / 2.0
    return np.array([1.0, h[0], h[1], h[2]])


def q_left(q):        # math_tools.h Qleft
    m = np.zeros((4, 4))
    m[0, 0] = q[0]; m[0, 1:] = -q[1:]; m[1:, 0] = q[1:]; m[1:, 1:] = q[0] * np.eye(3) + skew(q[1:])
    return m


def q_right(p):       # math_tools.h Qright
    m = np.zeros((4, 4))
    m[0, 0] = p[0]; m[0, 1:] = -p[1:]; m[1:, 0] = p[1:]; m[1:, 1:] = p[0] * np.eye(3) - skew(p[1:])
    return m


def left_quat_matrix(q):   # math_tools.h LeftQuatMatrix: (x, y, z, w) ordering
    m = np.zeros((4, 4))
    m[:3, :3] = q[0] * np.eye(3) + skew(q[1:]); m[3, :3] = -q[1:]; m[:3, 3] = q[1:]; m[3, 3] = q[0]
    return m


# ---------------------------------------------------------------- Preintegration.h
class Preintegration:
    def __init__(self, acc0, gyr0, ba, bg):
        self.acc0, self.gyr0 = np.array(acc0, np.float64), np.array(gyr0, np.float64)
        self.linearized_acc, self.linearized_gyr = self.acc0.copy(), self.gyr0.copy()
        self.ba, self.bg = np.array(ba, np.float64), np.array(bg, np.float64)
        self.jacobian = np.eye(15)
        self.sum_dt = 0.0
        self.delta_p, self.delta_q, self.delta_v = np.zeros(3), np.array([1.0, 0, 0, 0]), np.zeros(3)
        acc_n, gyr_n, acc_w, gyr_w = 0.00059, 0.000061, 0.000011, 0.000001                      # :39-42
        self.covariance = 0.0001 * np.eye(15)
        self.g_vec = -np.array([0.0, 0.0, 9.805])
        nz = np.zeros((18, 18))
        for k, s in ((0, acc_n), (3, gyr_n), (6, acc_n), (9, gyr_n), (12, acc_w), (15, gyr_w)):
            nz[k:k + 3, k:k + 3] = (s * s) * np.eye(3)
        self.noise = nz
        self.buf = []

    def push_back(self, dt, acc, gyr):                                                        # :57-62
        acc, gyr = np.array(acc, np.float64), np.array(gyr, np.float64)
        self.buf.append((dt, acc, gyr))
        self._propagate(dt, acc, gyr)

    def _propagate(self, dt, acc1, gyr1):                                                     # :163-184 with MidPointIntegration :79-161
        acc0, gyr0, dp, dq_, dv, ba, bg = self.acc0, self.gyr0, self.delta_p, self.delta_q, self.delta_v, self.ba, self.bg
        un_acc_0 = qrot(dq_, acc0 - ba)
        un_gyr = 0.5 * (gyr0 + gyr1) - bg
        rq = qmul(dq_, np.array([1.0, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2]))
        un_acc_1 = qrot(rq, acc1 - ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        w_x = 0.5 * (gyr0 + gyr1) - bg
        R_w_x, R_a_0_x, R_a_1_x = skew(w_x), skew(acc0 - ba), skew(acc1 - ba)
        Rd, Rr, I3 = qmat(dq_), qmat(rq), np.eye(3)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * Rd @ R_a_0_x * dt * dt + -0.25 * Rr @ R_a_1_x @ (I3 - R_w_x * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (Rd + Rr) * dt * dt
        F[0:3, 12:15] = -0.1667 * Rr @ R_a_1_x * dt * dt * -dt
        F[3:6, 3:6] = I3 - R_w_x * dt
        F[3:6, 12:15] = -I3 * dt
        F[6:9, 3:6] = -0.5 * Rd @ R_a_0_x * dt + -0.5 * Rr @ R_a_1_x @ (I3 - R_w_x * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (Rd + Rr) * dt
        F[6:9, 12:15] = -0.5 * Rr @ R_a_1_x * dt * -dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.5 * Rd * dt * dt
        V[0:3, 3:6] = -0.25 * Rr @ R_a_1_x * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.5 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * Rd * dt
        V[6:9, 3:6] = 0.5 * -Rr @ R_a_1_x * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        self.jacobian = F @ self.jacobian
        self.covariance = F @ self.covariance @ F.T + V @ self.noise @ V.T
        self.delta_p, self.delta_q, self.delta_v = rp, qnormalized(rq), rv
        self.sum_dt += dt
        self.acc0, self.gyr0 = acc1, gyr1

    def evaluate(self, Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj):                           # :186-225
        J = self.jacobian
        dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg = J[O_P:O_P + 3, O_BA:O_BA + 3], J[O_P:O_P + 3, O_BG:O_BG + 3], J[O_R:O_R + 3, O_BG:O_BG + 3], \
            J[O_V:O_V + 3, O_BA:O_BA + 3], J[O_V:O_V + 3, O_BG:O_BG + 3]
        dba, dbg = Bai - self.ba, Bgi - self.bg
        cq = qmul(self.delta_q, delta_q(dq_dbg @ dbg))
        cv = self.delta_v + dv_dba @ dba + dv_dbg @ dbg
        cp = self.delta_p + dp_dba @ dba + dp_dbg @ dbg
        s, g = self.sum_dt, self.g_vec
        r = np.zeros(15)
        Qi_inv = qinv(Qi)
        r[O_P:O_P + 3] = qrot(Qi_inv, -0.5 * g * s * s + Pj - Pi - Vi * s) - cp
        r[O_R:O_R + 3] = 2.0 * qnormalized(qmul(qinv(cq), qmul(Qi_inv, Qj)))[1:]
        r[O_V:O_V + 3] = qrot(Qi_inv, -g * s + Vj - Vi) - cv
        r[O_BA:O_BA + 3] = Baj - Bai
        r[O_BG:O_BG + 3] = Bgj - Bgi
        return r


def imu_factor(pre, Pi, Qi, SBi, Pj, Qj, SBj):
    """ImuFactor::Evaluate (ImuFactor.h:18-144): residual[15] and the six Jacobian blocks (15x3, 15x4, 15x9, 15x3, 15x4, 15x9)."""
    Pi, Pj, SBi, SBj = (np.asarray(a, np.float64) for a in (Pi, Pj, SBi, SBj))
    Qi, Qj = qnormalized(np.asarray(Qi, np.float64)), qnormalized(np.asarray(Qj, np.float64))
    Vi, Bai, Bgi, Vj, Baj, Bgj = SBi[0:3], SBi[3:6], SBi[6:9], SBj[0:3], SBj[3:6], SBj[6:9]
    residual = pre.evaluate(Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj)
    sqrt_info = np.linalg.cholesky(np.linalg.inv(pre.covariance)).T                          # LLT(cov^-1).matrixL().transpose()
    residual = sqrt_info @ residual
    s, g, J = pre.sum_dt, pre.g_vec, pre.jacobian
    dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg = J[O_P:O_P + 3, O_BA:O_BA + 3], J[O_P:O_P + 3, O_BG:O_BG + 3], J[O_R:O_R + 3, O_BG:O_BG + 3], \
        J[O_V:O_V + 3, O_BA:O_BA + 3], J[O_V:O_V + 3, O_BG:O_BG + 3]
    Ri_inv = qmat(qinv(Qi))
    I3 = np.eye(3)
    cq = qmul(pre.delta_q, delta_q(dq_dbg @ (Bgi - pre.bg)))
    u = Qi[1:]

    def dq_block(tmp):       # d(Qi^-1 tmp)/d(w | x y z) as the reference writes it (:63-64, :71-72)
        c0 = 2 * (Qi[0] * tmp + skew(u) @ tmp)
        c1 = 2 * (u.dot(tmp) * I3 + np.outer(u, tmp) - np.outer(tmp, u) - Qi[0] * skew(tmp))
        return c0, c1

    j0 = np.zeros((15, 3)); j0[O_P:O_P + 3, O_P:O_P + 3] = -Ri_inv
    j1 = np.zeros((15, 4))
    tmp = -0.5 * g * s * s + Pj - Pi - Vi * s
    c0, c1 = dq_block(tmp)
    j1[O_P:O_P + 3, 0] = c0; j1[O_P:O_P + 3, 1:4] = c1
    j1[O_R:O_R + 3, 0:4] = -2 * (q_left(qinv(Qj)) @ q_right(cq))[1:4, 0:4]
    tmp1 = -g * s + Vj - Vi
    c0, c1 = dq_block(tmp1)
    j1[O_V:O_V + 3, 0] = c0; j1[O_V:O_V + 3, 1:4] = c1
    j2 = np.zeros((15, 9))
    j2[O_P:O_P + 3, 0:3] = -Ri_inv * s
    j2[O_P:O_P + 3, 3:6] = -dp_dba
    j2[O_P:O_P + 3, 6:9] = -dp_dbg
    j2[O_R:O_R + 3, 6:9] = -left_quat_matrix(qmul(qmul(qinv(Qj), Qi), cq))[0:3, 0:3] @ dq_dbg
    j2[O_V:O_V + 3, 0:3] = -Ri_inv
    j2[O_V:O_V + 3, 3:6] = -dv_dba
    j2[O_V:O_V + 3, 6:9] = -dv_dbg
    j2[O_BA:O_BA + 3, 3:6] = -I3
    j2[O_BG:O_BG + 3, 6:9] = -I3
    j3 = np.zeros((15, 3)); j3[O_P:O_P + 3, O_P:O_P + 3] = Ri_inv
    j4 = np.zeros((15, 4)); j4[O_R:O_R + 3, 0:4] = 2 * q_left(qmul(qinv(cq), qinv(Qi)))[1:4, 0:4]
    j5 = np.zeros((15, 9))
    j5[O_V:O_V + 3, 0:3] = Ri_inv
    j5[O_BA:O_BA + 3, 3:6] = I3
    j5[O_BG:O_BG + 3, 6:9] = I3
    return residual, [sqrt_info @ j for j in (j0, j1, j2, j3, j4, j5)]


def speed_bias_prior(prior, sb):
    """SpeedBiasPriorFactorAutoDiff (PriorFactor.h:13-23): r = 15 (x - x0), J = 15 I."""
    return 15.0 * (np.asarray(sb, np.float64) - np.asarray(prior, np.float64)), [15.0 * np.eye(9)]


# ---------------------------------------------------------------- problem = parameter blocks + residual blocks
def plus_jacobian(q):     # ceres::QuaternionParameterization::ComputeJacobian, 4x3
    w, x, y, z = q
    return np.array([[-x, -y, -z], [w, z, -y], [-z, w, x], [y, -x, w]], dtype=np.float64)


def quat_plus(q, d):      # ceres::QuaternionParameterization::Plus
    nd = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    if nd > 0.0:
        sbd = np.sin(nd) / nd
        return qmul(np.array([np.cos(nd), sbd * d[0], sbd * d[1], sbd * d[2]]), q)
    return q.copy()


class Problem:
    """Parameter blocks by name -> (value array, 'quat' | None); residual blocks = (function(values...) -> (r, [J...]), block names, loss).
    A function may return `cost` as a third element (a block whose 1/2 |r|^2 is not its cost: the GPU Gram factor carries the padding
    row instead, so it never needs this)."""

    def __init__(self):
        self.params, self.kind, self.order, self.blocks = {}, {}, [], []

    def add_parameter(self, name, value, quat=False):
        self.params[name] = np.array(value, np.float64)
        self.kind[name] = "quat" if quat else None
        self.order.append(name)

    def add_residual(self, fn, names, loss=None):
        self.blocks.append((fn, list(names), loss))

    def local_sizes(self):
        return [3 if self.kind[n] else len(self.params[n]) for n in self.order]

    def evaluate(self, values=None, want_jac=True):
        """cost, residual vector (robustified), dense Jacobian in LOCAL coordinates."""
        values = values or self.params
        sizes = self.local_sizes()
        offs = dict(zip(self.order, np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(int)))
        rows, jrows, cost = [], [], 0.0
        for fn, names, loss in self.blocks:
            out = fn(*[values[n] for n in names])
            r, Js = np.atleast_1d(np.asarray(out[0], np.float64)), out[1]
            sq = float(r @ r)
            if len(out) == 3:                        # rows that are already robustified: the block hands over sum 1/2 rho itself
                cost += float(out[2])
            elif loss is not None:                     # Triggs corrector exactly as ceres::Corrector (same algebra as MarginalizationFactor.cpp:44-70)
                rho = loss(sq)
                cost += 0.5 * rho[0]
                sqrt_rho1 = np.sqrt(rho[1])
                if sq == 0.0 or rho[2] <= 0.0:
                    rs, alpha_sq_norm = sqrt_rho1, 0.0
                else:
                    D = 1.0 + 2.0 * sq * rho[2] / rho[1]
                    alpha = 1.0 - np.sqrt(D)
                    rs, alpha_sq_norm = sqrt_rho1 / (1 - alpha), alpha / sq
                Js = [sqrt_rho1 * (J - alpha_sq_norm * np.outer(r, r @ J)) for J in Js]
                r = r * rs
            else:
                cost += 0.5 * sq
            rows.append(r)
            if want_jac:
                Jd = np.zeros((len(r), int(sum(sizes))))
                for n, J in zip(names, Js):
                    J = np.asarray(J, np.float64).reshape(len(r), -1)
                    if self.kind[n]:
                        J = J @ plus_jacobian(values[n])
                    Jd[:, offs[n]:offs[n] + J.shape[1]] += J
                jrows.append(Jd)
        r = np.concatenate(rows)
        return cost, r, (np.vstack(jrows) if want_jac else None)

    def plus(self, values, delta):
        out, k = {}, 0
        for n, s in zip(self.order, self.local_sizes()):
            out[n] = quat_plus(values[n], delta[k:k + 3]) if self.kind[n] else values[n] + delta[k:k + s]
            k += s
        return out


def cauchy_loss(a):
    b, c = a * a, 1.0 / (a * a)

    def rho(s):
        sm = 1.0 + s * c
        inv = 1.0 / sm
        return b * np.log(sm), max(np.finfo(np.float64).tiny, inv), -c * (inv * inv)
    return rho


def ceres_lm(problem, max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8,
             initial_radius=1e4, max_radius=1e16, min_radius=1e-32, min_relative_decrease=1e-3, min_lm_diagonal=1e-6, max_lm_diagonal=1e32, log=None):
    """Ceres 2.0 TrustRegionMinimizer with LevenbergMarquardtStrategy and a dense linear solver, default options except
    max_num_iterations (L/src/BackendFusion.cpp:984-992 sets DENSE_QR and max_num_iter only).  Jacobi scaling on; monotonic steps."""
    x = {k: v.copy() for k, v in problem.params.items()}
    cost, r, J = problem.evaluate(x)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))               # jacobi_scaling
    radius, decrease = initial_radius, 2.0
    n_ok, n_invalid, termination = 0, 0, "max_iterations"
    for it in range(max_num_iterations):
        Js = J * scale
        g = Js.T @ r
        if np.abs(J.T @ r).max() <= gradient_tolerance:         # gradient max-norm (all blocks here are unconstrained)
            termination = "gradient_tolerance"
            break
        # LM step: min |Js d + r|^2 + |D d|^2, D^2 = clamp(diag(Js^T Js)) / radius
        diag = np.clip((Js * Js).sum(0), min_lm_diagonal, max_lm_diagonal)
        D = np.sqrt(diag / radius)
        A = np.vstack([Js, np.diag(D)])
        rhs = np.concatenate([-r, np.zeros(len(D))])
        d_s = np.linalg.lstsq(A, rhs, rcond=None)[0]            # DENSE_QR on the augmented system
        delta = d_s * scale
        model_change = -(d_s @ (g + 0.5 * (Js.T @ (Js @ d_s))))   # model_cost_change = -step^T (g + J^T J step / 2)
        if model_change <= 0:
            # invalid step (TrustRegionMinimizer::HandleInvalidStep): LevenbergMarquardtStrategy::StepIsInvalid halves the radius;
            # max_num_consecutive_invalid_steps (5) in a row -> FAILURE; MinTrustRegionRadiusReached -> CONVERGENCE
            n_invalid += 1
            radius *= 0.5
            if n_invalid >= 5:
                termination = "numerical_failure"
                break
            if radius <= min_radius and it + 1 < max_num_iterations:
                termination = "min_radius"
                break
            continue
        n_invalid = 0
        x_new = problem.plus(x, delta)
        new_cost = problem.evaluate(x_new, want_jac=False)[0]
        rho = (cost - new_cost) / model_change
        if log is not None:
            log.append(dict(it=it, cost=cost, new_cost=new_cost, rho=rho, radius=radius, step=float(np.linalg.norm(delta))))
        # Ceres checks both tolerances on the CANDIDATE, before it decides whether to take the step
        xnorm = np.sqrt(sum(float(v @ v) for v in x.values()))
        # (TrustRegionMinimizer::Minimize returns from ParameterToleranceReached / FunctionToleranceReached before IsStepSuccessful /
        # HandleSuccessfulStep: the candidate that triggers a tolerance is NOT taken — x, cost and the successful-step count stay)
        if float(np.linalg.norm(delta)) <= parameter_tolerance * (xnorm + parameter_tolerance):
            termination = "parameter_tolerance"
            break
        if abs(cost - new_cost) <= function_tolerance * cost:
            termination = "function_tolerance"
            break
        if rho > min_relative_decrease:                          # successful step
            x = x_new
            cost, r, J = problem.evaluate(x)
            radius = min(max_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            decrease = 2.0
            n_ok += 1
        else:
            radius = radius / decrease                           # LevenbergMarquardtStrategy::StepRejected (no clamp)
            decrease *= 2.0
            if radius <= min_radius and it + 1 < max_num_iterations:   # MinTrustRegionRadiusReached (checked after max_num_iterations)
                termination = "min_radius"
                break
    return x, dict(cost=cost, iterations=it + 1, successful_steps=n_ok, termination=termination)


# ---------------------------------------------------------------- marginalisation (MarginalizationFactor.cpp:128-287)
class Marginalization:
    """Dense restatement of MarginalizationInfo::Marginalize and MarginalizationFactor::Evaluate.  blocks: list of
    (name, value, is_quat) in the order [dropped ..., kept ...]; A, b = the information matrix / vector in LOCAL coordinates
    (quaternion blocks contribute their last three global columns, MarginalizationFactor.cpp:9-11 — callers pass them like that)."""
    EPS = 1e-8

    def __init__(self, A, b, m, kept):
        A, b = np.asarray(A, np.float64), np.asarray(b, np.float64)
        n = A.shape[0] - m
        Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
        w, V = np.linalg.eigh(Amm)
        Amm_inv = V @ np.diag(np.where(w > self.EPS, 1.0 / np.where(w > self.EPS, w, 1.0), 0.0)) @ V.T
        Amr, Arm, Arr = A[:m, m:], A[m:, :m], A[m:, m:]
        As = Arr - Arm @ Amm_inv @ Amr
        bs = b[m:] - Arm @ Amm_inv @ b[:m]
        w2, V2 = np.linalg.eigh(As)
        S = np.where(w2 > self.EPS, w2, 0.0)
        S_inv = np.where(w2 > self.EPS, 1.0 / np.where(w2 > self.EPS, w2, 1.0), 0.0)
        self.linearized_jacobians = np.diag(np.sqrt(S)) @ V2.T
        self.linearized_residuals = np.diag(np.sqrt(S_inv)) @ V2.T @ bs
        self.n, self.kept = n, [(name, np.array(val, np.float64), bool(q)) for name, val, q in kept]

    def factor(self):
        """Residual-block function over the kept parameter blocks (MarginalizationFactor::Evaluate)."""
        kept, LJ, LR, n = self.kept, self.linearized_jacobians, self.linearized_residuals, self.n

        def fn(*values):
            dx, idx, Js = np.zeros(n), 0, []
            for (name, x0, is_q), x in zip(kept, values):
                size = len(x0)
                if not is_q:
                    dx[idx:idx + size] = x - x0
                    Js.append(LJ[:, idx:idx + size])
                    idx += size
                else:
                    dq_ = qmul(qinv(x0), x)
                    sgn = 1.0 if dq_[0] >= 0 else -1.0
                    dx[idx:idx + 3] = sgn * 2.0 * qnormalized(dq_)[1:]
                    Js.append(sgn * 2.0 * LJ[:, idx:idx + 3] @ q_left(qinv(x0))[1:4, 0:4])
                    idx += 3
            return LR + LJ @ dx, Js
        return fn
