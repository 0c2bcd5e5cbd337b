/*
 * kirch_oracle.c -- plain-C CPU restatement of the reference's Kirchhoff
 * diffraction sum.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's
 * cpu_baseline leg); never linked into the product library.
 *
 * Follows src/impdar/lib/migrationlib/mig_python.py:35-60 (ImpDAR v1.2.1)
 * operation by operation in double precision, but visits only the traces
 * that can be inside the aperture instead of building the reference's
 * snum x tnum |tt - t| matrix per output sample (:49).  Pinned against the
 * golden vectors in tests/golden (tests/test_oracle_golden.py).
 *
 * Build: make -C oracle   ->  oracle/libkirch_oracle.so
 */
#include <math.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int kirch_oracle_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* nearest sample to t, ties to the lower index (argmin semantics, :49) */
static int pick(const double *tt, int n, double t, double tt0, double inv_dt)
{
    int k0 = (int)floor((t - tt0) * inv_dt);
    if (k0 < 0) k0 = 0;
    if (k0 > n - 1) k0 = n - 1;
    while (k0 < n - 1 && tt[k0 + 1] <= t) ++k0;
    while (k0 > 0 && tt[k0] > t) --k0;
    int k1 = k0 + 1 < n ? k0 + 1 : n - 1;
    return (fabs(tt[k1] - t) < fabs(tt[k0] - t)) ? k1 : k0;
}

/*
 * gradD, data: (snum, tnum) row-major float64 (data may be NULL unless
 * nearfield).  out: (snum, ntr) row-major, column c = output trace traces[c].
 * dist in metres, tt in seconds.
 */
void kirch_oracle(const double *gradD, const double *data, int snum, int tnum,
                  const double *dist, const double *tt, double vel, int nearfield,
                  const int *traces, int ntr, double *out)
{
    double tmax = tt[0];
    for (int k = 1; k < snum; ++k)
        if (tt[k] > tmax) tmax = tt[k];
    const double dt = snum > 1 ? (tt[snum - 1] - tt[0]) / (snum - 1) : 1.0;
    const double inv_dt = dt > 0 ? 1.0 / dt : 1.0;
    const double rlim = vel * tmax / 2.0;
    const double r2lim = rlim * rlim * (1.0 + 1e-9);
    const double c2pi = 1.0 / (2.0 * 3.141592653589793);
    int sorted = 1;
    for (int j = 1; j < tnum; ++j)
        if (!(dist[j] >= dist[j - 1])) sorted = 0;

#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < ntr; ++c) {
        const int xi = traces[c];
        const double dxi = dist[xi];
        /* trace range that can reach the aperture at all (sorted profiles) */
        int jlo = 0, jhi = tnum - 1;
        if (sorted) {
            while (jlo < xi && dxi - dist[jlo] > rlim * (1.0 + 1e-9)) ++jlo;
            while (jhi > xi && dist[jhi] - dxi > rlim * (1.0 + 1e-9)) --jhi;
        }
        for (int ti = 0; ti < snum; ++ti) {
            const double zs = vel * tt[ti] / 2.0;     /* :101 */
            const double zs2 = zs * zs;               /* :102 */
            double far = 0.0, near = 0.0;
            for (int j = jlo; j <= jhi; ++j) {
                const double dx = dist[j] - dxi;
                const double q = dx * dx + zs2;       /* :44 */
                if (q > r2lim) continue;
                const double rs = sqrt(q);
                const double cost = zs / rs;          /* :47 */
                const double t = 2.0 * rs / vel;      /* :49 */
                if (t > tmax) continue;               /* :52 */
                const int k = pick(tt, snum, t, tt[0], inv_dt);
                const double term = gradD[(size_t)k * tnum + j] * cost / vel;   /* :53 */
                if (term == term) far += term;
                if (nearfield) {
                    const double term2 = data[(size_t)k * tnum + j] * cost / (rs * rs);  /* :58 */
                    if (term2 == term2) near += term2;
                }
            }
            out[(size_t)ti * ntr + c] = c2pi * (far + near);   /* :60 */
        }
    }
}
