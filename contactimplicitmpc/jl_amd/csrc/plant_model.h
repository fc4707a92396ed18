// Plant-side residual of the reference's planar-chain models (quadruped, flamingo) for the batched simulator step
// (SURVEY.md section 8f-4).  Host- and device-compilable.
//
//   residual            src/simulation/simulation.jl:133-158     (LinearizedCone, flat ground: surface rotation = identity)
//   dynamics            src/dynamics/model.jl:11-36              (variational midpoint integrator)
//   quadruped           src/dynamics/quadruped/model.jl:75-590   flamingo  src/dynamics/flamingo/model.jl:62-503
//
// A model is a table: bodies and contact points are chains of (signed length, angle index) segments from the hip at
// (x, z) - a segment adds r (sin θ, -cos θ).  For such a chain with absolute angles the Lagrangian derivatives the
// integrator needs are explicit sums over the bodies,
//   D2L = dL/dq' ,   D1L = dL/dq - (d/dq dL/dq') q'      (dynamics/model.jl:11-15 with C of quadruped/model.jl:479-484),
// so no code generation is involved.  The Jacobian dr/dz comes from evaluating the same code on first-order dual
// numbers (one tangent direction per lane in the kernel): exact derivatives.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define PLANT_HD __host__ __device__ __forceinline__
#else
#define PLANT_HD inline
#endif

namespace cimpc {

struct Dual {            // value + one tangent, eps^2 = 0
    double v, d;
};
PLANT_HD Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
PLANT_HD Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
PLANT_HD Dual operator-(Dual a) { return {-a.v, -a.d}; }
PLANT_HD Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
PLANT_HD Dual operator*(double a, Dual b) { return {a * b.v, a * b.d}; }
PLANT_HD Dual operator*(Dual a, double b) { return {a.v * b, a.d * b}; }
PLANT_HD Dual operator+(Dual a, double b) { return {a.v + b, a.d}; }
PLANT_HD Dual operator-(Dual a, double b) { return {a.v - b, a.d}; }
PLANT_HD Dual operator/(Dual a, double b) { return {a.v / b, a.d / b}; }
PLANT_HD Dual psin(Dual a) { return {sin(a.v), a.d * cos(a.v)}; }
PLANT_HD Dual pcos(Dual a) { return {cos(a.v), -a.d * sin(a.v)}; }
PLANT_HD double psin(double a) { return sin(a); }
PLANT_HD double pcos(double a) { return cos(a); }
PLANT_HD double pval(double a) { return a; }
PLANT_HD double pval(Dual a) { return a.v; }
template <class T> PLANT_HD T pconst(double a);
template <> PLANT_HD double pconst<double>(double a) { return a; }
template <> PLANT_HD Dual pconst<Dual>(double a) { return {a, 0.0}; }

constexpr int PLANT_MAX_Q = 18, PLANT_MAX_U = 12, PLANT_MAX_BODIES = 9, PLANT_MAX_SEG = 3;
constexpr int PLANT_NC = 4, PLANT_NB = 16, PLANT_NW = 3;     // maxima: four contacts, two (flat_2D_lc) or four (flat_3D_lc) friction directions each
constexpr int PLANT_KIND_CHAIN = 0, PLANT_KIND_HOPPER_2D = 1, PLANT_KIND_CENTROIDAL = 2, PLANT_KIND_PARTICLE = 3;

struct PlantChain { int n; double r[PLANT_MAX_SEG]; int k[PLANT_MAX_SEG]; };
struct PlantModel {
    int kind = PLANT_KIND_CHAIN;      // planar chain with absolute angles | hopper_2D (constant mass matrix, prismatic leg)
    int nc = PLANT_NC;                // contacts
    int fd = 2;                       // friction directions per contact: 2 (planar) or 4 (spatial)
    int nw = 2;                       // disturbance dimension (A = eye(nw, nq))
    int nq, nu, n_bodies;
    double g, mu_world;
    double mass[PLANT_MAX_BODIES], inertia[PLANT_MAX_BODIES];
    int own[PLANT_MAX_BODIES];
    PlantChain body[PLANT_MAX_BODIES];
    PlantChain foot[PLANT_NC];
    int tq_a[PLANT_MAX_U], tq_b[PLANT_MAX_U];     // actuator i: torque between links a and b (B = -e_a + e_b)
    double joint_friction[PLANT_MAX_Q];
    PLANT_HD int nb() const { return fd * nc; }
    PLANT_HD int nz() const { return nq + 4 * nc + 2 * nb(); }
    PLANT_HD int nth() const { return 2 * nq + nu + nw + 2; }
};

// D1L, D2L at (q, v), accumulated into d1, d2 (nq entries each, zeroed here)
template <class T>
PLANT_HD void plant_lagrangian_derivatives(const PlantModel& M, const T* q, const T* v, T* d1, T* d2) {
    T s[PLANT_MAX_Q], c[PLANT_MAX_Q];
    if (M.kind == PLANT_KIND_HOPPER_2D) {
        // hopper_2D/model.jl:41-55: M = diag(mb + ml, mb + ml, Jb + Jl, ml) (kept in mass[0..3]), C = (0, (mb + ml) g, 0, 0);
        // lagrangian = 0, so D1L = -C and D2L = M v (dynamics/model.jl:11-15)
        for (int i = 0; i < 4; ++i) { d1[i] = pconst<T>(0.0); d2[i] = M.mass[i] * v[i]; }
        d1[1] = pconst<T>(-(M.mass[1] * M.g));
        return;
    }
    for (int i = 0; i < M.nq; ++i) { d1[i] = pconst<T>(0.0); d2[i] = pconst<T>(0.0); s[i] = psin(q[i]); c[i] = pcos(q[i]); }
    for (int b = 0; b < M.n_bodies; ++b) {
        const double m = M.mass[b];
        const PlantChain& ch = M.body[b];
        T vx = v[0], vz = v[1], ax = pconst<T>(0.0), az = pconst<T>(0.0);
        for (int e = 0; e < ch.n; ++e) {
            const int k = ch.k[e]; const double r = ch.r[e];
            vx = vx + r * (c[k] * v[k]);
            vz = vz + r * (s[k] * v[k]);
            ax = ax - r * (s[k] * (v[k] * v[k]));      // derivative of the body velocity along q' in q
            az = az + r * (c[k] * (v[k] * v[k]));
        }
        d2[0] = d2[0] + m * vx;
        d2[1] = d2[1] + m * vz;
        d1[0] = d1[0] - m * ax;
        d1[1] = d1[1] - (m * az + m * M.g);
        for (int e = 0; e < ch.n; ++e) {
            const int k = ch.k[e]; const double r = ch.r[e];
            d2[k] = d2[k] + (r * m) * (c[k] * vx + s[k] * vz);
            // dL/dθ_k - d(dL/dθ'_k)/dq q': the velocity-product terms cancel, gravity and the centripetal part stay
            d1[k] = d1[k] - ((m * M.g * r) * s[k] + (r * m) * (c[k] * ax + s[k] * az));
        }
        d2[M.own[b]] = d2[M.own[b]] + M.inertia[b] * v[M.own[b]];
    }
}

// ---- centroidal_quadruped (src/dynamics/centroidal_quadruped/model.jl): q = (body position, body orientation, four foot
// positions), one rigid body + point feet.  M = blkdiag(m_b I, I_b, m_f I_12) (:67-74), C = (m_b g e_z, w x I_b w, m_f g e_z ...)
// (:76-85) with lagrangian = 0, so D1L = -C, D2L = M v; B(q)^T u = (sum u_i, sum R^T skew(r_i) u_i, -u_1 .. -u_4) with the
// Euler rotation R of euler.jl:3-11 and r_i = foot_i - body (:98-121); contacts are the feet themselves (J = selection, :129-138),
// four friction directions per contact (flat_3D_lc: m = [1 0 -1 0; 0 1 0 -1]).  mass[0] = m_b, mass[1] = m_f, inertia[0..2] = I_b.
template <class T>
PLANT_HD void plant_centroidal_derivatives(const PlantModel& M, const T* v, T* d1, T* d2) {
    const double mb = M.mass[0], mf = M.mass[1];
    const double I0 = M.inertia[0], I1 = M.inertia[1], I2 = M.inertia[2];
    for (int i = 0; i < 3; ++i) { d2[i] = mb * v[i]; d1[i] = pconst<T>(i == 2 ? -mb * M.g : 0.0); }
    const T wx = v[3], wy = v[4], wz = v[5];
    d2[3] = I0 * wx; d2[4] = I1 * wy; d2[5] = I2 * wz;
    // -(w x I w)
    d1[3] = -(wy * (I2 * wz) - wz * (I1 * wy));
    d1[4] = -(wz * (I0 * wx) - wx * (I2 * wz));
    d1[5] = -(wx * (I1 * wy) - wy * (I0 * wx));
    for (int i = 6; i < 18; ++i) { d2[i] = mf * v[i]; d1[i] = pconst<T>((i % 3) == 2 ? -mf * M.g : 0.0); }
}
template <class T>
PLANT_HD void plant_residual_centroidal(const PlantModel& M, const T* z, const double* th, double kappa, T* r) {
    constexpr int nq = 18, nu = 12, nc = 4, nb = 16;
    const double* q0 = th; const double* q1 = th + nq; const double* u1 = th + 2 * nq; const double* w1 = u1 + nu;
    const double mu = w1[3], h = w1[4];
    const T* q2 = z; const T* gam = z + nq; const T* b = gam + nc; const T* psi = b + nb; const T* s1 = psi + nc;
    const T* eta = s1 + nc; const T* s2 = eta + nb;
    T qm2[nq], vm1[nq], vm2[nq];
    for (int i = 0; i < nq; ++i) { vm1[i] = pconst<T>((q1[i] - q0[i]) / h); qm2[i] = (q2[i] + q1[i]) * 0.5; vm2[i] = (q2[i] - q1[i]) / h; }
    T a1[nq], b1[nq], a2[nq], b2[nq];
    plant_centroidal_derivatives(M, vm1, a1, b1);
    plant_centroidal_derivatives(M, vm2, a2, b2);
    T dyn[nq];
    for (int i = 0; i < nq; ++i)
        dyn[i] = (0.5 * h) * a1[i] + b1[i] + (0.5 * h) * a2[i] - b2[i] - (h * M.joint_friction[i]) * vm2[i];
    {   // B(qm2)^T u
        const T sa = psin(qm2[3]), ca = pcos(qm2[3]), sb = psin(qm2[4]), cb = pcos(qm2[4]), sc = psin(qm2[5]), cc = pcos(qm2[5]);
        T R[3][3];
        R[0][0] = ca * cb; R[0][1] = ca * sb * sc - sa * cc; R[0][2] = ca * sb * cc + sa * sc;
        R[1][0] = sa * cb; R[1][1] = sa * sb * sc + ca * cc; R[1][2] = sa * sb * cc - ca * sc;
        R[2][0] = -sb;     R[2][1] = cb * sc;                R[2][2] = cb * cc;
        for (int f = 0; f < 4; ++f) {
            const double ux = u1[3 * f], uy = u1[3 * f + 1], uz = u1[3 * f + 2];
            const T rx = qm2[6 + 3 * f] - qm2[0], ry = qm2[7 + 3 * f] - qm2[1], rz = qm2[8 + 3 * f] - qm2[2];
            // skew(r) u = r x u
            const T cx = ry * uz - rz * uy, cy = rz * ux - rx * uz, cz = rx * uy - ry * ux;
            dyn[0] = dyn[0] + ux; dyn[1] = dyn[1] + uy; dyn[2] = dyn[2] + uz;
            for (int k = 0; k < 3; ++k) dyn[3 + k] = dyn[3 + k] + (R[0][k] * cx + R[1][k] * cy + R[2][k] * cz);      // R^T (r x u)
            dyn[6 + 3 * f] = dyn[6 + 3 * f] - ux; dyn[7 + 3 * f] = dyn[7 + 3 * f] - uy; dyn[8 + 3 * f] = dyn[8 + 3 * f] - uz;
        }
    }
    for (int i = 0; i < 3; ++i) dyn[i] = dyn[i] + w1[i];
    for (int f = 0; f < nc; ++f) {
        const T* bf = b + 4 * f; const T* ef = eta + 4 * f;
        // J^T lambda: the foot's own coordinates, lambda = [m b; gamma]
        dyn[6 + 3 * f] = dyn[6 + 3 * f] + (bf[0] - bf[2]);
        dyn[7 + 3 * f] = dyn[7 + 3 * f] + (bf[1] - bf[3]);
        dyn[8 + 3 * f] = dyn[8 + 3 * f] + gam[f];
        const T vx = (q2[6 + 3 * f] - q1[6 + 3 * f]) / h, vy = (q2[7 + 3 * f] - q1[7 + 3 * f]) / h;
        r[nq + f] = s1[f] - q2[8 + 3 * f];                           // s1 - phi(q2)
        r[nq + nc + 4 * f + 0] = ef[0] - vx - psi[f];                // eta - m^T v_T - E^T psi
        r[nq + nc + 4 * f + 1] = ef[1] - vy - psi[f];
        r[nq + nc + 4 * f + 2] = ef[2] + vx - psi[f];
        r[nq + nc + 4 * f + 3] = ef[3] + vy - psi[f];
        r[nq + nc + nb + f] = s2[f] - (mu * gam[f] - (bf[0] + bf[1] + bf[2] + bf[3]));
        r[nq + 2 * nc + nb + f] = gam[f] * s1[f] - kappa;
        for (int k = 0; k < 4; ++k) r[nq + 3 * nc + nb + 4 * f + k] = bf[k] * ef[k] - kappa;
        r[nq + 3 * nc + 2 * nb + f] = psi[f] * s2[f] - kappa;
    }
    for (int i = 0; i < nq; ++i) r[i] = dyn[i];
}

// ---- particle (src/dynamics/particle/model.jl): a point mass that is its own contact point; M = m I, C = (0, 0, m g),
// B = A = J = I, four friction directions (flat_3D_lc).  mass[0] = m.
template <class T>
PLANT_HD void plant_residual_particle(const PlantModel& M, const T* z, const double* th, double kappa, T* r) {
    const double* q0 = th; const double* q1 = th + 3; const double* u1 = th + 6; const double* w1 = th + 9;
    const double mu = th[12], h = th[13], m = M.mass[0];
    const T* q2 = z; const T* gam = z + 3; const T* b = z + 4; const T* psi = z + 8; const T* s1 = z + 9; const T* eta = z + 10; const T* s2 = z + 14;
    T vm2[3];
    for (int i = 0; i < 3; ++i) vm2[i] = (q2[i] - q1[i]) / h;
    const T lam[3] = {b[0] - b[2], b[1] - b[3], gam[0]};
    for (int i = 0; i < 3; ++i) {
        const double grav = i == 2 ? -m * M.g : 0.0;
        r[i] = pconst<T>(0.5 * h * grav + m * ((q1[i] - q0[i]) / h) + 0.5 * h * grav + u1[i] + w1[i]) - m * vm2[i] + lam[i];
    }
    r[3] = s1[0] - q2[2];
    r[4] = eta[0] - vm2[0] - psi[0]; r[5] = eta[1] - vm2[1] - psi[0];
    r[6] = eta[2] + vm2[0] - psi[0]; r[7] = eta[3] + vm2[1] - psi[0];
    r[8] = s2[0] - (mu * gam[0] - (b[0] + b[1] + b[2] + b[3]));
    r[9] = gam[0] * s1[0] - kappa;
    for (int k = 0; k < 4; ++k) r[10 + k] = b[k] * eta[k] - kappa;
    r[14] = psi[0] * s2[0] - kappa;
}

// r(z, θ, κ): z = [q2; γ; b; ψ; s1; η; s2], θ = [q0; q1; u1; w1; μ; h] (θ real: only dr/dz is needed)
template <class T>
PLANT_HD void plant_residual(const PlantModel& M, const T* z, const double* th, double kappa, T* r) {
    if (M.kind == PLANT_KIND_CENTROIDAL) { plant_residual_centroidal<T>(M, z, th, kappa, r); return; }
    if (M.kind == PLANT_KIND_PARTICLE) { plant_residual_particle<T>(M, z, th, kappa, r); return; }
    const int nq = M.nq, nu = M.nu, nc = M.nc, nb = M.nb();
    const double* q0 = th; const double* q1 = th + nq; const double* u1 = th + 2 * nq; const double* w1 = u1 + nu;
    const double mu = w1[M.nw], h = w1[M.nw + 1];
    const T* q2 = z; const T* gam = z + nq; const T* b = gam + nc; const T* psi = b + nb; const T* s1 = psi + nc;
    const T* eta = s1 + nc; const T* s2 = eta + nb;
    T qm1[PLANT_MAX_Q], vm1[PLANT_MAX_Q], qm2[PLANT_MAX_Q], vm2[PLANT_MAX_Q];
    for (int i = 0; i < nq; ++i) {
        qm1[i] = pconst<T>(0.5 * (q0[i] + q1[i])); vm1[i] = pconst<T>((q1[i] - q0[i]) / h);
        qm2[i] = (q2[i] + q1[i]) * 0.5; vm2[i] = (q2[i] - q1[i]) / h;
    }
    T a1[PLANT_MAX_Q], b1[PLANT_MAX_Q], a2[PLANT_MAX_Q], b2[PLANT_MAX_Q];
    plant_lagrangian_derivatives(M, qm1, vm1, a1, b1);
    plant_lagrangian_derivatives(M, qm2, vm2, a2, b2);
    T dyn[PLANT_MAX_Q];
    for (int i = 0; i < nq; ++i)
        dyn[i] = (0.5 * h) * a1[i] + b1[i] + (0.5 * h) * a2[i] - b2[i] - (h * M.joint_friction[i]) * vm2[i];
    if (M.kind == PLANT_KIND_HOPPER_2D) {
        // B(qm2)^T u, hopper_2D/model.jl:68-71: body torque on t; leg force along the leg axis (-sin t, cos t) and on r
        const T st = psin(qm2[2]), ct = pcos(qm2[2]);
        dyn[2] = dyn[2] + u1[0];
        dyn[0] = dyn[0] - u1[1] * st; dyn[1] = dyn[1] + u1[1] * ct; dyn[3] = dyn[3] + u1[1];
    } else {
        for (int i = 0; i < nu; ++i) { dyn[M.tq_a[i]] = dyn[M.tq_a[i]] - u1[i]; dyn[M.tq_b[i]] = dyn[M.tq_b[i]] + u1[i]; }
    }
    for (int i = 0; i < M.nw; ++i) dyn[i] = dyn[i] + w1[i];
    // contacts: position, Jacobian rows (x and z) of every foot at q2
    T s[PLANT_MAX_Q], c[PLANT_MAX_Q];
    for (int i = 0; i < nq; ++i) { s[i] = psin(q2[i]); c[i] = pcos(q2[i]); }
    for (int f = 0; f < nc; ++f) {
        const PlantChain& ch = M.foot[f];
        T pz = q2[1];
        T lx = b[2 * f] - b[2 * f + 1], lz = gam[f];                 // contact force [m b; γ]
        T vx = (q2[0] - q1[0]) / h;                                  // tangential foot velocity J_x (q2 - q1) / h
        dyn[0] = dyn[0] + lx; dyn[1] = dyn[1] + lz;                  // J^T λ: base columns
        if (M.kind == PLANT_KIND_HOPPER_2D) {
            // foot = (x + r sin t, z - r cos t), J = [1 0 r cos t sin t; 0 1 r sin t -cos t] at q2 (hopper_2D/model.jl:35-66)
            const T rl = q2[3];
            pz = pz - rl * c[2];
            dyn[2] = dyn[2] + rl * (c[2] * lx + s[2] * lz);
            dyn[3] = dyn[3] + (s[2] * lx - c[2] * lz);
            vx = vx + rl * (c[2] * ((q2[2] - q1[2]) / h)) + s[2] * ((q2[3] - q1[3]) / h);
        }
        for (int e = 0; e < (M.kind == PLANT_KIND_HOPPER_2D ? 0 : ch.n); ++e) {
            const int k = ch.k[e]; const double rr = ch.r[e];
            pz = pz - rr * c[k];
            dyn[k] = dyn[k] + rr * (c[k] * lx + s[k] * lz);
            vx = vx + rr * (c[k] * ((q2[k] - q1[k]) / h));
        }
        r[nq + f] = s1[f] - pz;                                      // s1 - ϕ(q2)
        r[nq + nc + 2 * f] = eta[2 * f] - vx - psi[f];               // η - v_T stack - Eᵀψ
        r[nq + nc + 2 * f + 1] = eta[2 * f + 1] + vx - psi[f];
        r[nq + nc + nb + f] = s2[f] - (mu * gam[f] - (b[2 * f] + b[2 * f + 1]));
        r[nq + 2 * nc + nb + f] = gam[f] * s1[f] - kappa;
        r[nq + 3 * nc + nb + 2 * f] = b[2 * f] * eta[2 * f] - kappa;
        r[nq + 3 * nc + nb + 2 * f + 1] = b[2 * f + 1] * eta[2 * f + 1] - kappa;
        r[nq + 3 * nc + 2 * nb + f] = psi[f] * s2[f] - kappa;
    }
    for (int i = 0; i < nq; ++i) r[i] = dyn[i];
}

// ---- the two models the reference tests in closed loop ---------------------------------------------------------------
inline PlantChain plant_chain(int n, double r0, int k0, double r1 = 0, int k1 = 0, double r2 = 0, int k2 = 0) {
    PlantChain c{}; c.n = n; c.r[0] = r0; c.k[0] = k0; c.r[1] = r1; c.k[1] = k1; c.r[2] = r2; c.k[2] = k2; return c;
}
inline PlantModel plant_quadruped() {          // quadruped/model.jl:516-575
    PlantModel M{};
    M.nq = 11; M.nu = 8; M.g = 9.81; M.mu_world = 1.0;
    const double m_torso = 4.713 + 4 * 0.696, m_thigh = 1.013, m_leg = 0.166;
    const double J_torso = 0.01683 + 4 * 0.696 * 0.183 * 0.183, J_thigh = 0.00552, J_leg = 0.00299;
    const double l_torso = 0.183 * 2, l_thigh = 0.2, l_leg = 0.2;
    const double d_torso = 0.5 * l_torso + 0.0127, d_thigh = 0.5 * l_thigh - 0.00323, d_leg = 0.5 * l_leg - 0.006435;
    int nb = 0;
    auto add = [&](double m, double J, int own, PlantChain c) { M.mass[nb] = m; M.inertia[nb] = J; M.own[nb] = own; M.body[nb] = c; ++nb; };
    add(m_torso, J_torso, 2, plant_chain(1, d_torso, 2));
    const int legs[4][2] = {{3, 4}, {5, 6}, {7, 8}, {9, 10}};
    for (int l = 0; l < 4; ++l) {
        const int th = legs[l][0], ca = legs[l][1];
        if (l < 2) {
            add(m_thigh, J_thigh, th, plant_chain(1, d_thigh, th));
            add(m_leg, J_leg, ca, plant_chain(2, l_thigh, th, d_leg, ca));
            M.foot[l] = plant_chain(2, l_thigh, th, l_leg, ca);
        } else {                                   // legs 3, 4 hang from the far end of the torso
            add(m_thigh, J_thigh, th, plant_chain(2, l_torso, 2, d_thigh, th));
            add(m_leg, J_leg, ca, plant_chain(3, l_torso, 2, l_thigh, th, d_leg, ca));
            M.foot[l] = plant_chain(3, l_torso, 2, l_thigh, th, l_leg, ca);
        }
    }
    M.n_bodies = nb;
    const int pairs[8][2] = {{2, 3}, {3, 4}, {2, 5}, {5, 6}, {2, 7}, {7, 8}, {2, 9}, {9, 10}};
    for (int i = 0; i < 8; ++i) { M.tq_a[i] = pairs[i][0]; M.tq_b[i] = pairs[i][1]; }
    for (int i = 0; i < 11; ++i) M.joint_friction[i] = i < 3 ? 0.0 : 0.1;
    return M;
}
inline PlantModel plant_hopper_2d() {          // hopper_2D/model.jl:95-121
    PlantModel M{};
    M.kind = PLANT_KIND_HOPPER_2D; M.nc = 1;
    M.nq = 4; M.nu = 2; M.g = 9.81; M.mu_world = 0.8; M.n_bodies = 0;
    const double mb = 3.0, ml = 0.3, Jb = 0.75, Jl = 0.075;
    M.mass[0] = mb + ml; M.mass[1] = mb + ml; M.mass[2] = Jb + Jl; M.mass[3] = ml;      // diagonal of the mass matrix
    for (int i = 0; i < 4; ++i) M.joint_friction[i] = 0.0;
    return M;
}
inline PlantModel plant_centroidal(bool damped) {          // centroidal_quadruped/model.jl:187-228
    PlantModel M{};
    M.kind = PLANT_KIND_CENTROIDAL; M.nc = 4; M.fd = 4; M.nw = 3;
    M.nq = 18; M.nu = 12; M.g = 9.81; M.mu_world = 0.3; M.n_bodies = 0;
    M.mass[0] = 13.5; M.mass[1] = 0.2;
    M.inertia[0] = 0.0178533 * 10.0; M.inertia[1] = 0.0377999 * 10.0; M.inertia[2] = 0.0456542 * 10.0;
    for (int i = 0; i < 18; ++i) M.joint_friction[i] = damped ? ((i >= 3 && i < 6) ? 30.0 : 10.0) : 0.0;      // mu_joint = 1
    return M;
}
inline PlantModel plant_particle() {           // particle/model.jl:113-121
    PlantModel M{};
    M.kind = PLANT_KIND_PARTICLE; M.nc = 1; M.fd = 4; M.nw = 3;
    M.nq = 3; M.nu = 3; M.g = 9.81; M.mu_world = 1.0; M.n_bodies = 0;
    M.mass[0] = 1.0;
    for (int i = 0; i < 3; ++i) M.joint_friction[i] = 0.0;
    return M;
}
inline PlantModel plant_flamingo() {           // flamingo/model.jl:458-495
    PlantModel M{};
    M.nq = 9; M.nu = 6; M.g = 9.81; M.mu_world = 0.9;
    const double m_torso = 12.0, m_thigh = 0.4598, m_calf = 0.306, m_foot = 0.3466;
    const double J_torso = 0.10, J_thigh = 0.01256, J_calf = 0.00952, J_foot = 0.0015;
    const double l_thigh = 0.42, l_calf = 0.45, l_foot = 0.1725;
    const double d_torso = 0.20, d_thigh = 0.21, d_calf = 0.225, d_foot = 0.0525, cb = 0.5 * (l_foot - d_foot);
    int nb = 0;
    auto add = [&](double m, double J, int own, PlantChain c) { M.mass[nb] = m; M.inertia[nb] = J; M.own[nb] = own; M.body[nb] = c; ++nb; };
    add(m_torso, J_torso, 2, plant_chain(1, -d_torso, 2));           // the torso points up from the hip
    const int legs[2][3] = {{3, 4, 7}, {5, 6, 8}};
    for (int l = 0; l < 2; ++l) {
        const int th = legs[l][0], ca = legs[l][1], ft = legs[l][2];
        add(m_thigh, J_thigh, th, plant_chain(1, d_thigh, th));
        add(m_calf, J_calf, ca, plant_chain(2, l_thigh, th, d_calf, ca));
        add(m_foot, J_foot, ft, plant_chain(3, l_thigh, th, l_calf, ca, cb, ft));
        M.foot[2 * l] = plant_chain(3, l_thigh, th, l_calf, ca, l_foot, ft);          // toe
        M.foot[2 * l + 1] = plant_chain(3, l_thigh, th, l_calf, ca, -d_foot, ft);     // heel
    }
    M.n_bodies = nb;
    const int pairs[6][2] = {{2, 3}, {3, 4}, {2, 5}, {5, 6}, {4, 7}, {6, 8}};
    for (int i = 0; i < 6; ++i) { M.tq_a[i] = pairs[i][0]; M.tq_b[i] = pairs[i][1]; }
    for (int i = 0; i < 9; ++i) M.joint_friction[i] = 0.0;
    return M;
}

}  // namespace cimpc
