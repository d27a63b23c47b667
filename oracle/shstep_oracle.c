/*
 * shstep_oracle.c — CPU restatement of docs/SPEC.md Part II: rigid-body properties of a shape (§5),
 * the nve integrator and body forces (§6), periodic ghosts and the half neighbour list (§7).
 *
 * TEST INFRASTRUCTURE ONLY, PARITY UNPINNED — same status and rules as shpair_oracle.c (see its
 * header): the reference's fix / neighbour sources are absent from the mount
 * (/root/reference/README.md:1 is the whole reference), so this restates the SPEC in plain scalar C
 * and is pinned by closed-form answers in tests/test_oracle_step.py (sphere and ellipsoid mass
 * properties, ballistic flight, torque-free tops, brute-force neighbour sets).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SHO_PI 3.14159265358979323846264338327950288

double sho_sh_eval(int L, const double *anm, const double u[3], double *grad);
void sho_gauss_legendre(int n, double *t, double *w);

/* ---------------------------------------------------------------- SPEC §5 */

/* out[10] = V, c[3], J_c (xx,yy,zz,xy,xz,yz), unit density, body frame. */
void sho_mass_props(int L, const double *anm, double out[10])
{
  const int nt = (5 * L) / 2 + 3, np = 5 * L + 4;
  double *t = malloc(sizeof(double) * nt), *w = malloc(sizeof(double) * nt);
  sho_gauss_legendre(nt, t, w);
  double V = 0, c[3] = {0, 0, 0}, J[6] = {0, 0, 0, 0, 0, 0};
  for (int a = 0; a < nt; ++a) {
    const double z = t[a], s = sqrt(fmax(0.0, 1.0 - z * z));
    for (int b = 0; b < np; ++b) {
      const double ph = 2.0 * SHO_PI * (b + 0.5) / np;
      const double u[3] = {s * cos(ph), s * sin(ph), z};
      const double r = sho_sh_eval(L, anm, u, NULL);
      const double dw = w[a] * 2.0 * SHO_PI / np;
      const double r3 = r * r * r, r4 = r3 * r, r5 = r4 * r;
      V += dw * r3 / 3.0;
      for (int k = 0; k < 3; ++k) c[k] += dw * r4 / 4.0 * u[k];
      J[0] += dw * r5 / 5.0 * (1.0 - u[0] * u[0]);
      J[1] += dw * r5 / 5.0 * (1.0 - u[1] * u[1]);
      J[2] += dw * r5 / 5.0 * (1.0 - u[2] * u[2]);
      J[3] -= dw * r5 / 5.0 * u[0] * u[1];
      J[4] -= dw * r5 / 5.0 * u[0] * u[2];
      J[5] -= dw * r5 / 5.0 * u[1] * u[2];
    }
  }
  for (int k = 0; k < 3; ++k) c[k] /= V;
  const double c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  out[0] = V;
  out[1] = c[0]; out[2] = c[1]; out[3] = c[2];
  out[4] = J[0] - V * (c2 - c[0] * c[0]);
  out[5] = J[1] - V * (c2 - c[1] * c[1]);
  out[6] = J[2] - V * (c2 - c[2] * c[2]);
  out[7] = J[3] + V * c[0] * c[1];
  out[8] = J[4] + V * c[0] * c[2];
  out[9] = J[5] + V * c[1] * c[2];
  free(t);
  free(w);
}

/* ---------------------------------------------------------------- SPEC §6 */

static void q2m(const double q[4], double R[3][3])
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = w * w + x * x - y * y - z * z; R[0][1] = 2 * (x * y - w * z); R[0][2] = 2 * (x * z + w * y);
  R[1][0] = 2 * (x * y + w * z); R[1][1] = w * w - x * x + y * y - z * z; R[1][2] = 2 * (y * z - w * x);
  R[2][0] = 2 * (x * z - w * y); R[2][1] = 2 * (y * z + w * x); R[2][2] = w * w - x * x - y * y + z * z;
}
static void mv(const double R[3][3], const double a[3], double o[3])
{
  for (int k = 0; k < 3; ++k) o[k] = R[k][0] * a[0] + R[k][1] * a[1] + R[k][2] * a[2];
}
static void mtv(const double R[3][3], const double a[3], double o[3])
{
  for (int k = 0; k < 3; ++k) o[k] = R[0][k] * a[0] + R[1][k] * a[1] + R[2][k] * a[2];
}
static void cross(const double a[3], const double b[3], double o[3])
{
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
/* inverse of the symmetric 3x3 (xx,yy,zz,xy,xz,yz) scaled by rho */
static void inertia_inverse(const double mp[10], double rho, double Ii[3][3])
{
  const double a = rho * mp[4], b = rho * mp[5], c = rho * mp[6], d = rho * mp[7], e = rho * mp[8], f = rho * mp[9];
  const double det = a * (b * c - f * f) - d * (d * c - f * e) + e * (d * f - b * e);
  Ii[0][0] = (b * c - f * f) / det; Ii[0][1] = (e * f - d * c) / det; Ii[0][2] = (d * f - b * e) / det;
  Ii[1][0] = Ii[0][1]; Ii[1][1] = (a * c - e * e) / det; Ii[1][2] = (d * e - a * f) / det;
  Ii[2][0] = Ii[0][2]; Ii[2][1] = Ii[1][2]; Ii[2][2] = (a * b - d * d) / det;
}
static void omega_of(const double q[4], const double L[3], const double Ii[3][3], double w[3])
{
  double R[3][3], lb[3], wb[3];
  q2m(q, R);
  mtv(R, L, lb);
  mv(Ii, lb, wb);
  mv(R, wb, w);
}
static void qnorm(double q[4])
{
  const double n = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] *= n;
}
/* qd = 1/2 (0, w) (x) q */
static void qdot(const double q[4], const double L[3], const double Ii[3][3], double qd[4])
{
  double w[3];
  omega_of(q, L, Ii, w);
  qd[0] = 0.5 * (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]);
  qd[1] = 0.5 * (w[0] * q[0] + w[1] * q[3] - w[2] * q[2]);
  qd[2] = 0.5 * (w[1] * q[0] + w[2] * q[1] - w[0] * q[3]);
  qd[3] = 0.5 * (w[2] * q[0] + w[0] * q[2] - w[1] * q[1]);
}
static void richardson(double q[4], const double L[3], const double Ii[3][3], double dt)
{
  double qd[4], qf[4], qh[4];
  qdot(q, L, Ii, qd);
  for (int k = 0; k < 4; ++k) { qf[k] = q[k] + dt * qd[k]; qh[k] = q[k] + 0.5 * dt * qd[k]; }
  qnorm(qf);
  qnorm(qh);
  qdot(qh, L, Ii, qd);
  for (int k = 0; k < 4; ++k) qh[k] += 0.5 * dt * qd[k];
  qnorm(qh);
  for (int k = 0; k < 4; ++k) q[k] = 2.0 * qh[k] - qf[k];
  qnorm(q);
}

/* massprops: nshape x 10 (sho_mass_props), density: nshape. phase 0 = initial_integrate, 1 = final. */
void sho_nve(int phase, int n, double dt, const double *massprops, const double *density, double *x, double *v,
             double *quat, double *angmom, const double *f, const double *torque, const int *shtype,
             const int *mask, int groupbit)
{
  for (int i = 0; i < n; ++i) {
    if (!(mask[i] & groupbit)) continue;
    const double *mp = massprops + 10 * shtype[i];
    const double rho = density[shtype[i]], m = rho * mp[0];
    double R[3][3], s[3], sf[3], Ii[3][3];
    double *q = quat + 4 * i, *L = angmom + 3 * i;
    q2m(q, R);
    mv(R, mp + 1, s);
    cross(s, f + 3 * i, sf);
    for (int k = 0; k < 3; ++k) {
      v[3 * i + k] += 0.5 * dt / m * f[3 * i + k];
      L[k] += 0.5 * dt * (torque[3 * i + k] - sf[k]);
    }
    if (phase == 1) continue;
    double X[3];
    for (int k = 0; k < 3; ++k) X[k] = x[3 * i + k] + s[k] + dt * v[3 * i + k];
    inertia_inverse(mp, rho, Ii);
    richardson(q, L, Ii, dt);
    q2m(q, R);
    mv(R, mp + 1, s);
    for (int k = 0; k < 3; ++k) x[3 * i + k] = X[k] - s[k];
  }
}

/* f += m g - gt v ; torque += s x F_b - gr w */
void sho_post_force(int n, const double *massprops, const double *density, const double g[3], double gamma_t,
                    double gamma_r, const double *v, const double *quat, const double *angmom, const int *shtype,
                    const int *mask, int groupbit, double *f, double *torque)
{
  for (int i = 0; i < n; ++i) {
    if (!(mask[i] & groupbit)) continue;
    const double *mp = massprops + 10 * shtype[i];
    const double rho = density[shtype[i]], m = rho * mp[0];
    double R[3][3], s[3], Fb[3], sF[3], Ii[3][3], w[3];
    q2m(quat + 4 * i, R);
    mv(R, mp + 1, s);
    for (int k = 0; k < 3; ++k) Fb[k] = m * g[k] - gamma_t * v[3 * i + k];
    cross(s, Fb, sF);
    inertia_inverse(mp, rho, Ii);
    omega_of(quat + 4 * i, angmom + 3 * i, Ii, w);
    for (int k = 0; k < 3; ++k) {
      f[3 * i + k] += Fb[k];
      torque[3 * i + k] += sF[k] - gamma_r * w[k];
    }
  }
}

/* out[0] = sum 1/2 m v^2, out[1] = sum 1/2 w.L, out[2] = sum -m g.(x+s) */
void sho_energies(int n, const double *massprops, const double *density, const double g[3], const double *x,
                  const double *v, const double *quat, const double *angmom, const int *shtype, const int *mask,
                  int groupbit, double out[3])
{
  out[0] = out[1] = out[2] = 0.0;
  for (int i = 0; i < n; ++i) {
    if (!(mask[i] & groupbit)) continue;
    const double *mp = massprops + 10 * shtype[i];
    const double rho = density[shtype[i]], m = rho * mp[0];
    double R[3][3], s[3], Ii[3][3], w[3];
    q2m(quat + 4 * i, R);
    mv(R, mp + 1, s);
    inertia_inverse(mp, rho, Ii);
    omega_of(quat + 4 * i, angmom + 3 * i, Ii, w);
    for (int k = 0; k < 3; ++k) {
      out[0] += 0.5 * m * v[3 * i + k] * v[3 * i + k];
      out[1] += 0.5 * w[k] * angmom[3 * i + k];
      out[2] -= m * g[k] * (x[3 * i + k] + s[k]);
    }
  }
}

/* ---------------------------------------------------------------- SPEC §7 */

/* Wraps owned x into the box and lists the periodic ghosts. ghost_owner / ghost_shift (3 ints each) must
 * hold 26 n entries at most; returns the number of ghosts. */
int sho_borders(int n, double *x, const double lo[3], const double hi[3], const int periodic[3], double cmax,
                int *ghost_owner, int *ghost_shift)
{
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d)
      if (periodic[d]) {
        const double len = hi[d] - lo[d];
        double *p = x + 3 * i + d;
        while (*p < lo[d]) *p += len;
        while (*p >= hi[d]) *p -= len;
      }
  int ng = 0;
  for (int i = 0; i < n; ++i)
    for (int code = 0; code < 27; ++code) {
      const int s[3] = {code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1};
      if (code == 13) continue;
      int ok = 1;
      for (int d = 0; d < 3; ++d) {
        if (s[d] && !periodic[d]) ok = 0;
        if (s[d] == 1 && !(x[3 * i + d] < lo[d] + cmax)) ok = 0;
        if (s[d] == -1 && !(x[3 * i + d] >= hi[d] - cmax)) ok = 0;
      }
      if (!ok) continue;
      ghost_owner[ng] = i;
      for (int d = 0; d < 3; ++d) ghost_shift[3 * ng + d] = s[d];
      ++ng;
    }
  return ng;
}

/* Brute-force half list over nall = nlocal + nghost particles. First call with jlist == NULL to size it.
 * offsets[nlocal+1]; returns the number of pairs. */
int sho_half_list(int nlocal, int nall, const double *x, const int *shtype, const int *tag, const double *rmax,
                  double skin, int *offsets, int *jlist)
{
  int np = 0;
  for (int i = 0; i < nlocal; ++i) {
    offsets[i] = np;
    for (int j = 0; j < nall; ++j) {
      if (!(tag[i] < tag[j])) continue;
      const double dx = x[3 * i] - x[3 * j], dy = x[3 * i + 1] - x[3 * j + 1], dz = x[3 * i + 2] - x[3 * j + 2];
      const double c = rmax[shtype[i]] + rmax[shtype[j]] + skin;
      if (dx * dx + dy * dy + dz * dz < c * c) {
        if (jlist) jlist[np] = j;
        ++np;
      }
    }
  }
  offsets[nlocal] = np;
  return np;
}
