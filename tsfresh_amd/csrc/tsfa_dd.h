// Double-double arithmetic (~32 significant digits: an unevaluated sum hi + lo of two float64) for the second passes
// that redo ill-conditioned least-squares fits (fam_ar_dd.h: AutoReg / adfuller, fam_langevin_dd.h: np.polyfit).
// Error-free transformations (two_sum, two_prod by fma); compiled with -ffp-contract=off like everything else.
#ifndef TSFA_DD_H
#define TSFA_DD_H

#include "tsfa_common.h"

struct dd {
    double hi, lo;
};
TSFA_DEV dd dd_from(double a) { return dd{a, 0.0}; }
TSFA_DEV dd dd_quick_two_sum(double a, double b) {
    const double s = a + b;
    return dd{s, b - (s - a)};
}
TSFA_DEV dd dd_two_sum(double a, double b) {
    const double s = a + b;
    const double bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
TSFA_DEV dd dd_two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, __builtin_fma(a, b, -p)};
}
TSFA_DEV dd dd_add(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi);
    const dd t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = dd_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return dd_quick_two_sum(s.hi, s.lo);
}
TSFA_DEV dd dd_neg(dd a) { return dd{-a.hi, -a.lo}; }
TSFA_DEV dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
TSFA_DEV dd dd_mul(dd a, dd b) {
    dd p = dd_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return dd_quick_two_sum(p.hi, p.lo);
}
TSFA_DEV dd dd_mul_d(dd a, double b) {
    dd p = dd_two_prod(a.hi, b);
    p.lo += a.lo * b;
    return dd_quick_two_sum(p.hi, p.lo);
}
TSFA_DEV dd dd_add_prod(dd acc, double a, double b) { return dd_add(acc, dd_two_prod(a, b)); }  // acc + a * b
TSFA_DEV dd dd_div(dd a, dd b) {
    const double q1 = a.hi / b.hi;
    dd r = dd_sub(a, dd_mul_d(b, q1));
    const double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul_d(b, q2));
    const double q3 = r.hi / b.hi;
    return dd_add(dd_quick_two_sum(q1, q2), dd_from(q3));
}
TSFA_DEV dd dd_sqrt(dd a) {
    if (!(a.hi > 0.0)) return dd_from(0.0);
    const double x = 1.0 / sqrt(a.hi);
    const double ax = a.hi * x;
    const dd t = dd_sub(a, dd_two_prod(ax, ax));
    return dd_two_sum(ax, t.hi * (x * 0.5));
}

#endif
