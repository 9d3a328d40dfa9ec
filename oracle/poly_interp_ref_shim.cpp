// poly_interp_ref_shim.cpp -- C entry points around the REFERENCE's own PolyQuintic
// (library/src/poly_interp.cu, compiled as host C++ from where it lies under /root/reference by
// oracle/Makefile into oracle/_ref/libpoly_interp_ref.so).  TEST INFRASTRUCTURE ONLY: used by
// tests/test_poly_lat_oracle.py to pin oracle/poly_lat_oracle.c's quintic and the product's
// host quintic against the reference implementation itself.
#include "tplcpp/poly_interp.cuh"

extern "C" {

__attribute__((visibility("default")))
void ref_quintic_init(double* state9, double x0, double y0, double dy0, double ddy0,
                      double x1, double y1, double dy1, double ddy1)
{
    PolyQuintic q(x0, y0, dy0, ddy0, x1, y1, dy1, ddy1);
    state9[0] = q.x0_; state9[1] = q.x1_; state9[2] = q.d;
    for (int i = 0; i < 6; ++i) state9[3 + i] = q.c[i];
}

__attribute__((visibility("default")))
double ref_quintic_eval(const double* state9, int order, double x)
{
    PolyQuintic q;
    q.x0_ = state9[0]; q.x1_ = state9[1]; q.d = state9[2];
    for (int i = 0; i < 6; ++i) q.c[i] = state9[3 + i];
    switch (order) {
    case 0: return q.f(x);
    case 1: return q.df(x);
    case 2: return q.ddf(x);
    default: return q.dddf(x);
    }
}

}
