/* cycle_window_model.c -- CPU what-if model (round 5): how soon after its orbit has settled does a pixel of the set retire
 * under the cycle test (csrc/mbk_loops.inc, MBK_G_CYC), as a function of WHEN the reference state is replaced?
 *
 * The kernels compare the state bitwise with a saved one every 8 steps and replace the saved state after `win` checks; round
 * 2-4 doubled `win` every time (Brent): the saved states sit at steps 8, 16, 32, 64, ... so a pixel that becomes bitwise
 * periodic at step s retires at the next power of two (x 8) above s plus lcm(8, period) -- on average 1.44 s.  Any
 * replacement schedule is exact (a bitwise repeat proves periodicity whatever the two steps are); a schedule whose windows grow
 * more slowly places a saved state sooner after s, at the price of catching long periods later (the window must span
 * lcm(8, period) steps).  This program runs the reference loop per pixel with the schedule
 *        win <- win + max(1, win >> shift)          (shift 0 = doubling, 1 = x1.5, 2 = x1.25, 3 = x1.125)
 * and prints, per view, the pixel-steps executed and the wave-steps of one-wave-per-8x8-block kernels (a wave runs until its
 * last lane has escaped or retired), next to the floor (retire at the first check at or after the step at which the orbit
 * became periodic: found by running Floyd/Brent on the full orbit).
 *
 *   gcc -O2 -fopenmp -ffp-contract=off -o /tmp/cwm scripts/cycle_window_model.c -lm && /tmp/cwm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void axis(double start, double range, uint32_t n, double *x)
{
    const double stop = start + range, delta = stop - start;
    if (n == 1) { x[0] = start; return; }
    const double div = (double)(n - 1), step = delta / div;
    for (uint32_t k = 0; k < n; ++k) x[k] = (double)k * step + start;
    x[n - 1] = stop;
}

/* executed steps under the schedule; count via *cnt (0 = never) */
static int32_t run(double cr, double ci, int32_t mrd, int32_t first, int shift, int32_t *cnt)
{
    double zr = cr, zi = ci, sr = 0, si = 0;
    int have = 0;
    uint32_t tc = 0, win = 1;
    for (int32_t n = 1; n < mrd; ++n) {
        const double a = zr * zr, b = zi * zi, t = a - b, w = 2.0 * zr, u = w * zi;
        zr = t + cr;
        zi = u + ci;
        if (zr * zr + zi * zi >= 4.0) { *cnt = n; return n; }
        if (n >= first && (n - first) % 8 == 0) {
            if (!have) { sr = zr; si = zi; have = 1; }
            else {
                if (memcmp(&zr, &sr, 8) == 0 && memcmp(&zi, &si, 8) == 0) { *cnt = 0; return n; }
                if (++tc >= win) {
                    sr = zr; si = zi; tc = 0;
                    if (shift < 0) win += 1u;                       /* linear */
                    else { const uint32_t inc = win >> shift; win += inc ? inc : 1u; }
                }
            }
        }
    }
    *cnt = 0;
    return mrd > 1 ? mrd - 1 : 0;
}

/* Alternating scheme: the grouped loops keep the state of 8 steps ago in registers anyway (the group's start state, which the
 * deferred replay needs), so every OTHER check can compare with THAT state instead of the saved one -- same two compares, no
 * save -- and catches every period that divides 8 at the first such check after the orbit has settled; the checks in between
 * (16 steps apart) run the saved-state scheme with the given window growth for all other periods. */
static int32_t run_alt(double cr, double ci, int32_t mrd, int32_t first, int shift, int32_t *cnt)
{
    double zr = cr, zi = ci, sr = 0, si = 0, pr = 0, pi = 0;
    int have = 0;
    uint32_t tc = 0, win = 1, k = 0;
    for (int32_t n = 1; n < mrd; ++n) {
        const double a = zr * zr, b = zi * zi, t = a - b, w = 2.0 * zr, u = w * zi;
        zr = t + cr;
        zi = u + ci;
        if (zr * zr + zi * zi >= 4.0) { *cnt = n; return n; }
        if (n >= first && (n - first) % 8 == 0) {
            if (!have) { sr = zr; si = zi; have = 1; k = 0; }
            else {
                ++k;
                if (k & 1u) {   /* against the state of 8 steps ago */
                    if (memcmp(&zr, &pr, 8) == 0 && memcmp(&zi, &pi, 8) == 0) { *cnt = 0; return n; }
                } else {
                    if (memcmp(&zr, &sr, 8) == 0 && memcmp(&zi, &si, 8) == 0) { *cnt = 0; return n; }
                    if (++tc >= win) {
                        sr = zr; si = zi; tc = 0;
                        if (shift < 0) win += 1u;
                        else { const uint32_t inc = win >> shift; win += inc ? inc : 1u; }
                    }
                }
            }
            pr = zr; pi = zi;
        }
    }
    *cnt = 0;
    return mrd > 1 ? mrd - 1 : 0;
}

/* Both at every check (priced, not built): compare with the state of 8 steps ago AND with the saved one -- periods 1, 2, 4, 8
 * retire at the first check after the orbit has settled, the rest under the product's schedule (a quarter's growth below `wcap`
 * checks, doubling above).  Costs two more 64-bit compares and two moves per 8 steps (+1.5-2 % instructions). */
static int32_t run_both(double cr, double ci, int32_t mrd, int32_t first, uint32_t wcap, int with_prev, int32_t *cnt)
{
    double zr = cr, zi = ci, sr = 0, si = 0, pr = 0, pi = 0;
    int have = 0;
    uint32_t tc = 0, win = 1;
    for (int32_t n = 1; n < mrd; ++n) {
        const double a = zr * zr, b = zi * zi, t = a - b, w = 2.0 * zr, u = w * zi;
        zr = t + cr;
        zi = u + ci;
        if (zr * zr + zi * zi >= 4.0) { *cnt = n; return n; }
        if (n >= first && (n - first) % 8 == 0) {
            if (!have) { sr = zr; si = zi; have = 1; }
            else {
                if (with_prev && memcmp(&zr, &pr, 8) == 0 && memcmp(&zi, &pi, 8) == 0) { *cnt = 0; return n; }
                if (memcmp(&zr, &sr, 8) == 0 && memcmp(&zi, &si, 8) == 0) { *cnt = 0; return n; }
                if (++tc >= win) {
                    sr = zr; si = zi; tc = 0;
                    win += win < wcap ? (win >> 2) + 1u : win;   /* the product's MBK_G_CYC_OOL */
                }
            }
            pr = zr; pi = zi;
        }
    }
    *cnt = 0;
    return mrd > 1 ? mrd - 1 : 0;
}

/* floor: the first check step (n >= first, (n - first) % 8 == 0) at or after BOTH the orbit's entry into its bitwise cycle and
 * one full lcm(8, period) later (a match needs two states lcm apart) -- what an oracle that knew the period could do */
static int32_t run_floor(double cr, double ci, int32_t mrd, int32_t first, double *hr, double *hi)
{
    double zr = cr, zi = ci;
    int32_t n;
    hr[0] = zr; hi[0] = zi;
    for (n = 1; n < mrd; ++n) {
        const double a = zr * zr, b = zi * zi, t = a - b, w = 2.0 * zr, u = w * zi;
        zr = t + cr;
        zi = u + ci;
        if (zr * zr + zi * zi >= 4.0) return n;
        hr[n] = zr; hi[n] = zi;
    }
    /* never escaped within mrd - 1 steps: find the smallest period p and entry step mu of the bitwise cycle, if any, inside the orbit */
    const int32_t last = mrd - 1;
    int32_t p = 0;
    for (int32_t q = 1; q <= last / 2; ++q)
        if (memcmp(&hr[last], &hr[last - q], 8) == 0 && memcmp(&hi[last], &hi[last - q], 8) == 0) { p = q; break; }
    if (!p) return last;
    int32_t mu = last - p;
    while (mu > 0 && memcmp(&hr[mu - 1], &hr[mu - 1 + p], 8) == 0 && memcmp(&hi[mu - 1], &hi[mu - 1 + p], 8) == 0) --mu;
    /* lcm(8, p) */
    int32_t g = 8, r = p;
    while (r) { const int32_t tt = g % r; g = r; r = tt; }
    const int32_t l = 8 / g * p;
    int32_t s = mu < first ? first : mu;
    s = first + ((s - first + 7) / 8) * 8;   /* first check at or after mu */
    s += l;
    return s < last ? s : last;
}

int main(int argc, char **argv)
{
    struct { const char *name; double sr, si, rr, ri; uint32_t n; int32_t mrd; } views[] = {
        {"cfg2 4096^2 mrd 1000", -2.0, -1.5, 3.0, 3.0, 4096, 1000},
        {"DataChunk (1,0,0) mrd 1000", -2.0, -2.0, 4.0, 4.0, 4096, 1000},
        {"DataChunk (1,0,0) mrd 1024", -2.0, -2.0, 4.0, 4.0, 4096, 1024},
        {"DataChunk (4,1,1) mrd 1024", -1.0, -1.0, 1.0, 1.0, 4096, 1024},
    };
    const int nviews = argc > 1 ? atoi(argv[1]) : 4;
    const int shifts[] = {0, 1, 2, 3, -1, 100, 101, 102, 200, 201};   /* 100 + s: the alternating scheme with shift s; 200 / 201: the product's schedule (cap 32) without / with the state of 8 steps ago */
    for (int v = 0; v < nviews && v < 4; ++v) {
        const uint32_t N = views[v].n;
        const int32_t mrd = views[v].mrd;
        double *xr = malloc(sizeof(double) * N), *xi = malloc(sizeof(double) * N);
        axis(views[v].sr, views[v].rr, N, xr);
        axis(views[v].si, views[v].ri, N, xi);
        const uint32_t nb = N / 8;
        printf("== %s\n", views[v].name);
        for (int pol = -1; pol < 10; ++pol) {
            double px_steps = 0, wave_steps = 0;
            long long never = 0, early = 0;
#pragma omp parallel for schedule(dynamic, 2) reduction(+ : px_steps, wave_steps, never, early)
            for (int64_t by = 0; by < (int64_t)nb; ++by) {
                double *hr = NULL, *hi = NULL;
                if (pol < 0) { hr = malloc(sizeof(double) * (size_t)mrd); hi = malloc(sizeof(double) * (size_t)mrd); }
                for (uint32_t bx = 0; bx < nb; ++bx) {
                    int32_t longest = 0;
                    /* the kernels' per-step prologue: 8 steps for the blocks that are not classified interior, 0 for those that
                     * are; modelled as 8 throughout (first check at step 8, then every 8) */
                    for (uint32_t ly = 0; ly < 8; ++ly)
                        for (uint32_t lx = 0; lx < 8; ++lx) {
                            int32_t cnt = 0, ex;
                            if (pol < 0) {
                                ex = run_floor(xr[bx * 8 + lx], xi[by * 8 + ly], mrd, 8, hr, hi);
                            } else {
                                ex = shifts[pol] >= 200 ? run_both(xr[bx * 8 + lx], xi[by * 8 + ly], mrd, 8, 32u, shifts[pol] - 200, &cnt)
                                   : shifts[pol] >= 100 ? run_alt(xr[bx * 8 + lx], xi[by * 8 + ly], mrd, 8, shifts[pol] - 100, &cnt)
                                                        : run(xr[bx * 8 + lx], xi[by * 8 + ly], mrd, 8, shifts[pol], &cnt);
                                if (cnt == 0) { ++never; if (ex < mrd - 1) ++early; }
                            }
                            px_steps += ex;
                            longest = ex > longest ? ex : longest;
                        }
                    wave_steps += longest;
                }
                free(hr); free(hi);
            }
            if (pol < 0)
                printf("  floor (period known)          pixel-steps %8.1f M  wave-steps %7.3f M  lane activity %.3f\n", px_steps / 1e6, wave_steps / 1e6,
                       px_steps / 64.0 / wave_steps);
            else if (shifts[pol] >= 200)
                printf("  product's schedule (cap 32)%s  pixel-steps %8.1f M  wave-steps %7.3f M  lane activity %.3f   never %lld, retired early %.1f %%\n",
                       shifts[pol] == 201 ? " + state of 8 steps ago" : "                       ", px_steps / 1e6, wave_steps / 1e6, px_steps / 64.0 / wave_steps, never,
                       never ? 100.0 * early / never : 0.0);
            else
                printf("  %s win += max(1, win >> %2d)%s  pixel-steps %8.1f M  wave-steps %7.3f M  lane activity %.3f   never %lld, retired early %.1f %%\n",
                       shifts[pol] >= 100 ? "alternating," : "            ", shifts[pol] % 100 > 50 ? -1 : shifts[pol] % 100, shifts[pol] == 0 ? " (r2-4)" : "       ", px_steps / 1e6, wave_steps / 1e6, px_steps / 64.0 / wave_steps, never,
                       never ? 100.0 * early / never : 0.0);
            fflush(stdout);
        }
        free(xr); free(xi);
    }
    return 0;
}
