/* c_abi_demo.c -- the drop-in boundary used from plain C99 (no C++, no torch): the reference's first known
 * answer (src/ell.rs:247-256: Ell::new_with_scalar(0.01, zeros(4)), g = 0.5 * ones(4), central cut) through
 * include/ellhip.h, one batched call through include/ellhip_batch.h and one LDLT through include/ellhip_lmi.h.
 * Also proves that the four public headers are valid C.  Prints "ok" and exits 0 on success. */
#include <math.h>
#include <stdio.h>

#include "../../include/ellhip.h"
#include "../../include/ellhip_batch.h"
#include "../../include/ellhip_lmi.h"
#include "../../include/ellhip_lowpass.h"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "FAILED %s (%s)\n", #cond, ellhip_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void) {
    if (ellhip_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", ellhip_version());
        return 2;
    }
    /* ---- Ell, one central cut */
    ellhip_space *h = NULL;
    double xc0[4] = {0, 0, 0, 0}, g[4] = {0.5, 0.5, 0.5, 0.5}, xc[4], mq[16];
    CHECK(ellhip_create(&h, ELLHIP_SPACE_ELL, 4, 0.01, NULL, NULL, xc0, -1) == 0);
    CHECK(ellhip_update(h, ELLHIP_CUT_CENTRAL, g, 0.0, 0, 0.0) == ELLHIP_SUCCESS);
    CHECK(ellhip_get_xc(h, xc) == 0 && ellhip_get_mq(h, mq) == 0);
    for (int i = 0; i < 4; ++i) CHECK(xc[i] == -0.01); /* bit-exact in the reference too */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) CHECK(mq[4 * i + j] == (i == j ? 1.0 : 0.0) - 0.1);
    CHECK(fabs(ellhip_kappa(h) - 0.16 / 15.0) < 1e-15 && fabs(ellhip_tsq(h) - 0.01) < 1e-15);
    ellhip_destroy(h);
    /* ---- three copies of the same problem in one batched call */
    ellhip_batch *b = NULL;
    double kap[3] = {0.01, 0.01, 0.01}, gb[12], b0[3] = {0, 0, 0}, b1[3] = {0, 0, 0}, xb[12];
    int32_t kinds[3] = {ELLHIP_CUT_CENTRAL, ELLHIP_CUT_CENTRAL, ELLHIP_CUT_CENTRAL}, has1[3] = {0, 0, 0}, st[3];
    for (int i = 0; i < 12; ++i) gb[i] = 0.5;
    CHECK(ellhip_batch_create(&b, 3, 4, kap, NULL, NULL, NULL, -1) == 0);
    CHECK(ellhip_batch_update(b, 1, kinds, gb, b0, has1, b1, st, NULL) == 0);
    CHECK(ellhip_batch_get_xc(b, xb) == 0);
    for (int i = 0; i < 12; ++i) CHECK(xb[i] == -0.01 && st[i / 4] == ELLHIP_SUCCESS);
    ellhip_batch_destroy(b);
    /* ---- LDLTMgr::factorize on chol1 (src/oracles/ldlt_mgr.rs:159-190) */
    ellhip_lmi *l = NULL;
    double chol1[9] = {25, 15, -5, 15, 18, 0, -5, 0, 11};
    int64_t pos[2];
    CHECK(ellhip_lmi_create(&l, 0, 3, NULL, chol1, -1) == 0);
    CHECK(ellhip_lmi_assess_feas(l, NULL, NULL, NULL) == 0); /* positive definite */
    CHECK(ellhip_lmi_pos(l, pos) == 0 && pos[1] == 0);
    ellhip_lmi_destroy(l);
    printf("ok\n");
    return 0;
}
