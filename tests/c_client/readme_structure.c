/* Plain-C client of include/dto_engine.h: builds the README problem's description (README.md:70-92 of the
 * reference: x[2], u[1], dt; G = [-0.1 1; -1 -0.1] + u [0 1; 1 0]; QuadraticRegularizer(:u, 1)) with a structure-only
 * handle (device = -1: no GPU needed), queries sizes and the sparsity structure, and prints them for the test to
 * compare with the oracle.  It is what the Julia shim does through @ccall, written in the ABI's own language. */
#include <stdio.h>
#include <stdlib.h>
#include "dto_engine.h"

int main(void) {
    enum { N = 50, Z = 4 };
    static double G[2 * 4] = {-0.1, -1.0, 1.0, -0.1, /* column-major G(0) */ 0.0, 1.0, 1.0, 0.0 /* drive */};
    static double Z0[N * Z];
    for (int k = 0; k < N; ++k) { Z0[k * Z + 0] = 1.0; Z0[k * Z + 1] = 0.0; Z0[k * Z + 2] = 0.1; Z0[k * Z + 3] = 0.1; }
    double R[1] = {1.0};
    dto_integrator_desc integ = {DTO_INTEGRATOR_BILINEAR, 0, 2, 2, 1, G};
    dto_objective_desc obj = {0};
    obj.kind = DTO_OBJECTIVE_QUADRATIC_REGULARIZER; obj.comp_off = 2; obj.comp_dim = 1; obj.weight = 1.0; obj.R = R;
    dto_problem_desc d = {0};
    d.abi_version = DTO_ABI_VERSION; d.device = -1; d.N = N; d.z = Z; d.gd = 0; d.dt_idx = 3; d.eval_hessian = 1;
    d.n_integrators = 1; d.n_objectives = 1; d.n_constraints = 0;
    d.integrators = &integ; d.objectives = &obj; d.constraints = NULL; d.Z0 = Z0;
    dto_handle* h = NULL;
    if (dto_create(&d, &h) != 0) { fprintf(stderr, "create: %s\n", dto_last_error(NULL)); return 1; }
    int64_t nv, nc, nd, jn, hn;
    dto_num_vars(h, &nv); dto_num_cons(h, &nc); dto_num_dynamics_cons(h, &nd); dto_jac_nnz(h, &jn); dto_hess_nnz(h, &hn);
    printf("sizes %lld %lld %lld %lld %lld\n", (long long)nv, (long long)nc, (long long)nd, (long long)jn, (long long)hn);
    int64_t* r = malloc(sizeof(int64_t) * (size_t)jn);
    int64_t* c = malloc(sizeof(int64_t) * (size_t)jn);
    if (dto_jacobian_structure(h, 0, jn, r, c) != 0) return 2;
    long long sr = 0, sc = 0;
    for (int64_t i = 0; i < jn; ++i) { sr += r[i] * (i % 7 + 1); sc += c[i] * (i % 5 + 1); }
    printf("jac_checksum %lld %lld first %lld %lld last %lld %lld\n", sr, sc, (long long)r[0], (long long)c[0],
           (long long)r[jn - 1], (long long)c[jn - 1]);
    int64_t hr[8], hc[8];
    if (dto_hessian_structure(h, hn - 8, 8, hr, hc) != 0) return 3;
    printf("hess_tail");
    for (int i = 0; i < 8; ++i) printf(" %lld:%lld", (long long)hr[i], (long long)hc[i]);
    printf("\n");
    /* evaluation needs a device: a structure-only handle must refuse loudly, not fall back */
    double f = 0.0;
    int rc = dto_eval_objective(h, Z0, &f);
    printf("eval_rc %d msg %s\n", rc, rc ? dto_last_error(h) : "none");
    dto_destroy(h);
    free(r); free(c);
    return 0;
}
