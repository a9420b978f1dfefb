// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/refshim/README.md).
//
// Stand-in for the part of Ceres Solver 2.0 that include/factors/LidarKeyframeFactor.h of the reference
// touches: ceres::Jet<T, N> (value + N partials, arithmetic as published in ceres/jet.h: product rule,
// quotient via the reciprocal of g.a, sqrt via 1 / (2 sqrt a)), ceres::CostFunction's Evaluate contract
//  and ceres::AutoDiffCostFunction<Functor, kNumResiduals, N0, N1>, which seeds one Jet per
// parameter scalar and copies the partials out row-major.  Loss functions are given for the losses
// the reference names (closed forms from ceres/loss_function.h).
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <string>
#include <vector>

namespace ceres {

template <class T, int N> struct Jet {
    T a; T v[N];
    Jet() : a(), v() {}
    Jet(const T& value) : a(value), v() {}   // NOLINT: implicit like ceres
    Jet(const T& value, int k) : a(value), v() { v[k] = T(1); }
    Jet& operator+=(const Jet& g) { *this = *this + g; return *this; }
    Jet& operator-=(const Jet& g) { *this = *this - g; return *this; }
    Jet& operator*=(const Jet& g) { *this = *this * g; return *this; }
    Jet& operator/=(const Jet& g) { *this = *this / g; return *this; }
};
template <class T, int N> Jet<T, N> operator+(const Jet<T, N>& f) { return f; }
template <class T, int N> Jet<T, N> operator-(const Jet<T, N>& f) { Jet<T, N> r; r.a = -f.a; for (int i = 0; i < N; i++) r.v[i] = -f.v[i]; return r; }
template <class T, int N> Jet<T, N> operator+(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a + g.a; for (int i = 0; i < N; i++) r.v[i] = f.v[i] + g.v[i]; return r; }
template <class T, int N> Jet<T, N> operator-(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a - g.a; for (int i = 0; i < N; i++) r.v[i] = f.v[i] - g.v[i]; return r; }
template <class T, int N> Jet<T, N> operator*(const Jet<T, N>& f, const Jet<T, N>& g) { Jet<T, N> r; r.a = f.a * g.a; for (int i = 0; i < N; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
template <class T, int N> Jet<T, N> operator/(const Jet<T, N>& f, const Jet<T, N>& g) {
    const T g_a_inverse = T(1.0) / g.a;
    const T f_a_by_g_a = f.a * g_a_inverse;
    Jet<T, N> r; r.a = f_a_by_g_a;
    for (int i = 0; i < N; i++) r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
    return r;
}
template <class T, int N> Jet<T, N> operator+(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a = f.a + s; return r; }
template <class T, int N> Jet<T, N> operator+(T s, const Jet<T, N>& f) { Jet<T, N> r = f; r.a = s + f.a; return r; }
template <class T, int N> Jet<T, N> operator-(const Jet<T, N>& f, T s) { Jet<T, N> r = f; r.a = f.a - s; return r; }
template <class T, int N> Jet<T, N> operator-(T s, const Jet<T, N>& f) { Jet<T, N> r; r.a = s - f.a; for (int i = 0; i < N; i++) r.v[i] = -f.v[i]; return r; }
template <class T, int N> Jet<T, N> operator*(const Jet<T, N>& f, T s) { Jet<T, N> r; r.a = f.a * s; for (int i = 0; i < N; i++) r.v[i] = f.v[i] * s; return r; }
template <class T, int N> Jet<T, N> operator*(T s, const Jet<T, N>& f) { Jet<T, N> r; r.a = f.a * s; for (int i = 0; i < N; i++) r.v[i] = f.v[i] * s; return r; }
template <class T, int N> Jet<T, N> operator/(const Jet<T, N>& f, T s) { const T inv = T(1.0) / s; Jet<T, N> r; r.a = f.a * inv; for (int i = 0; i < N; i++) r.v[i] = f.v[i] * inv; return r; }
template <class T, int N> bool operator<(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a < g.a; }
template <class T, int N> bool operator>(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a > g.a; }
template <class T, int N> bool operator>=(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a >= g.a; }
template <class T, int N> bool operator<=(const Jet<T, N>& f, const Jet<T, N>& g) { return f.a <= g.a; }
template <class T, int N> Jet<T, N> sqrt(const Jet<T, N>& f) {
    const T tmp = std::sqrt(f.a);
    const T two_a_inverse = T(1.0) / (T(2.0) * tmp);
    Jet<T, N> r; r.a = tmp; for (int i = 0; i < N; i++) r.v[i] = f.v[i] * two_a_inverse;
    return r;
}
template <class T, int N> Jet<T, N> abs(const Jet<T, N>& f) { return f.a < T(0) ? -f : f; }

class CostFunction {
public:
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
    const std::vector<int>& parameter_block_sizes() const { return sizes_; }
    int num_residuals() const { return nres_; }
protected:
    std::vector<int>* mutable_parameter_block_sizes() { return &sizes_; }      // as in ceres/cost_function.h (dynamically sized cost functions)
    void set_num_residuals(int n) { nres_ = n; }
    std::vector<int> sizes_; int nres_ = 0;
};

template <int kNumResiduals, int... Ns> class SizedCostFunction : public CostFunction {
public:
    SizedCostFunction() { sizes_ = {Ns...}; nres_ = kNumResiduals; }
};

template <class Functor, int kNumResiduals, int N0, int N1> class AutoDiffCostFunction : public CostFunction {
public:
    explicit AutoDiffCostFunction(Functor* f) : f_(f) { sizes_ = {N0, N1}; nres_ = kNumResiduals; }
    ~AutoDiffCostFunction() override { delete f_; }
    bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
        static_assert(kNumResiduals == 1, "the lidar factors have one residual");
        if (!jacobians) return (*f_)(p[0], p[1], residuals);
        typedef Jet<double, N0 + N1> J;
        J x0[N0], x1[N1], out[kNumResiduals];
        for (int i = 0; i < N0; i++) x0[i] = J(p[0][i], i);
        for (int i = 0; i < N1; i++) x1[i] = J(p[1][i], N0 + i);
        if (!(*f_)(x0, x1, out)) return false;
        residuals[0] = out[0].a;
        if (jacobians[0]) for (int i = 0; i < N0; i++) jacobians[0][i] = out[0].v[i];
        if (jacobians[1]) for (int i = 0; i < N1; i++) jacobians[1][i] = out[0].v[N0 + i];
        return true;
    }
    const Functor* functor() const { return f_; }   // shim-only accessor: lets the driver read the record a block was built from
private:
    Functor* f_;
};

// ceres/loss_function.h closed forms: rho[0] = rho(s), rho[1] = rho'(s), rho[2] = rho''(s)
class LossFunction { public: virtual ~LossFunction() {} virtual void Evaluate(double s, double rho[3]) const = 0; };
class CauchyLoss : public LossFunction {
public:
    explicit CauchyLoss(double a) : b_(a * a), c_(1 / b_) {}
    void Evaluate(double s, double rho[3]) const override {
        const double sum = 1.0 + s * c_; const double inv = 1.0 / sum;
        rho[0] = b_ * std::log(sum); rho[1] = std::max(std::numeric_limits<double>::min(), inv); rho[2] = -c_ * (inv * inv);
    }
private: const double b_, c_;
};
class HuberLoss : public LossFunction {
public:
    explicit HuberLoss(double a) : a_(a), b_(a * a) {}
    void Evaluate(double s, double rho[3]) const override {
        if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s); }
        else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    }
private: const double a_, b_;
};


// ---- ceres::Problem / ceres::Solve surface --------------------------------------------------------------------
// The reference's nodes build a Problem (AddParameterBlock / AddResidualBlock) and call ceres::Solve.  The Ceres
// trust-region solver itself is NOT restated: Solve() hands the problem to a hook the driver installs, which records the
// residual blocks the reference created (that is the product of the reference's own code) and writes back whatever
// step the test protocol prescribes.
struct LocalParameterization { virtual ~LocalParameterization() {} };
struct QuaternionParameterization : LocalParameterization {};
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
struct ResidualBlock { CostFunction* cost; LossFunction* loss; std::vector<double*> params; };
class Problem {
public:
    std::vector<ResidualBlock> blocks;
    std::vector<LocalParameterization*> parameterizations;
    void AddParameterBlock(double*, int, LocalParameterization* p = nullptr) { if (p) parameterizations.push_back(p); }
    template <class... P> void AddResidualBlock(CostFunction* c, LossFunction* l, P*... ps) { blocks.push_back(ResidualBlock{c, l, {ps...}}); }
    ~Problem() {   // ceres::Problem owns cost functions, loss functions and parameterizations (each deleted once)
        std::vector<const void*> seen;
        auto once = [&](const void* p) { if (!p || std::find(seen.begin(), seen.end(), p) != seen.end()) return false; seen.push_back(p); return true; };
        for (auto& b : blocks) { if (once(b.cost)) delete b.cost; }
        for (auto& b : blocks) { if (once(b.loss)) delete b.loss; }
        for (auto* p : parameterizations) if (once(p)) delete p;
    }
};
struct Solver {
    struct Options {
        LinearSolverType linear_solver_type = DENSE_QR;
        TrustRegionStrategyType trust_region_strategy_type = LEVENBERG_MARQUARDT;
        int max_num_iterations = 50, num_threads = 1;
        double max_solver_time_in_seconds = 1e9, gradient_check_relative_precision = 1e-8;
        bool minimizer_progress_to_stdout = false, check_gradients = false;
    };
    struct Summary { int num_blocks = 0; std::string BriefReport() const { return "refshim"; } std::string FullReport() const { return "refshim"; } };
};
typedef void (*SolveHook)(const Solver::Options&, Problem*, Solver::Summary*);
inline SolveHook& solve_hook() { static SolveHook h = nullptr; return h; }
inline void Solve(const Solver::Options& o, Problem* p, Solver::Summary* s) { if (solve_hook()) solve_hook()(o, p, s); }

}  // namespace ceres
