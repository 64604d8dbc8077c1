/* Cubic-Hermite activation tables of the round-1 experiment: included by the PATCHED TEMPORARY COPY of
 * oracle/p3d_oracle.c that tools/experiments/table_activation_error.py builds.  Never part of the shipped oracle. */
#define OR_TAB_N 320            /* intervals of width 1/16 on |x| in [0, 20) */
static float or_tab_sp[OR_TAB_N + 1][4], or_tab_sg[OR_TAB_N + 1][4];
__attribute__((constructor)) static void or_tab_init(void) {
    const double h = 1.0 / 16.0;
    for (int i = 0; i <= OR_TAB_N; ++i) {
        double u0 = i * h, u1 = u0 + h;
        /* g(u) = log1p(exp(-u)), g' = -1/(1+exp(u));  s(u) = 1/(1+exp(-u)), s' = s(1-s) */
        double g0 = log1p(exp(-u0)), g1 = log1p(exp(-u1)), dg0 = -1.0 / (1.0 + exp(u0)), dg1 = -1.0 / (1.0 + exp(u1));
        double s0 = 1.0 / (1.0 + exp(-u0)), s1 = 1.0 / (1.0 + exp(-u1)), ds0 = s0 * (1 - s0), ds1 = s1 * (1 - s1);
        if (i == OR_TAB_N) { g0 = g1 = dg0 = dg1 = 0.0; s0 = s1 = 1.0; ds0 = ds1 = 0.0; }
        or_tab_sp[i][0] = (float)g0; or_tab_sp[i][1] = (float)(h * dg0);
        or_tab_sp[i][2] = (float)(3 * (g1 - g0) - h * (2 * dg0 + dg1)); or_tab_sp[i][3] = (float)(2 * (g0 - g1) + h * (dg0 + dg1));
        or_tab_sg[i][0] = (float)s0; or_tab_sg[i][1] = (float)(h * ds0);
        or_tab_sg[i][2] = (float)(3 * (s1 - s0) - h * (2 * ds0 + ds1)); or_tab_sg[i][3] = (float)(2 * (s0 - s1) + h * (ds0 + ds1));
    }
}
static inline float or_tab_eval(const float (*tab)[4], float u) {
    float t = fminf(u, (float)OR_TAB_N / 16.0f) * 16.0f; /* exact scaling */
    int i = (int)t;
    float f = t - (float)i;
    const float* c = tab[i];
    return fmaf(fmaf(fmaf(c[3], f, c[2]), f, c[1]), f, c[0]);
}
static inline float or_softplus_h(float x) { return fmaxf(x, 0.0f) + or_tab_eval(or_tab_sp, fabsf(x)); }
static inline float or_sigmoid_c(float x) {
    float s = or_tab_eval(or_tab_sg, fabsf(x));
    return x >= 0.0f ? s : 1.0f - s;
}
