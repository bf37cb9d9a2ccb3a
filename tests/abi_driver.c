/* abi_driver.c -- the C ABI of include/cmf_hip.h consumed from plain C99, the way Julia's `ccall` consumes it:
 * prototypes from the header only, column-major Float64 buffers, status codes + cmf_last_error().
 * TEST INFRASTRUCTURE (built by tests/test_abi_driver.py with `gcc -std=c99 -Wall -Werror -I include`).
 *
 *   abi_driver <in.bin> <out.bin> [ndev] [rule]
 *   rule: mult (default) | gram (mult with cmf_set_option(h, "gram", 1)) | hals | pgd
 *
 * in.bin  : int64 N, T, K, L, iters; double l1W, l2W, l1H, l2H; data[N*T], W0[K*N*L], H0[K*T]   (Julia memory order)
 * out.bin : double loss[iters + 1] (compute_loss, then update_feature_maps! per iteration), W[K*N*L], H[K*T],
 *           then the same three again from cmf_fit on a fresh handle (loss_hist, W, H)
 * The sequence is the reference's: MultUpdate(data, W, H) (mult.jl:11-20, model.jl:79), compute_loss
 * (alternating.jl:37), then update_motifs! / update_feature_maps! per iteration (alternating.jl:52,54).
 * With ndev > 0 the rule is the T-sharded group on devices 0..ndev-1 (all 0 when only one GPU exists).
 * hals: HALSUpdate (hals.jl:18-42) = cmf_create + cmf_set_factors + "hals_prepare", then cmf_hals_update_motifs /
 * cmf_hals_update_feature_maps; pgd: PGDUpdate (pgd.jl:112-202) with its defaults -- loss_func = SquareLoss(), constrW =
 * constrH = NonnegConstraint(), penaltiesW = [SquarePenalty(1)], penaltiesH = [] -- through cmf_pgd_reset /
 * cmf_pgd_update_motifs / cmf_pgd_update_feature_maps.  Those two rules have no cmf_fit: the second pass repeats the calls. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cmf_hip.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != CMF_OK) {                                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cmf_last_error());              \
            return 10 + rc_;                                                              \
        }                                                                                 \
    } while (0)

static int read_all(FILE *f, void *p, size_t bytes) { return fread(p, 1, bytes, f) == bytes ? 0 : 1; }

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.bin out.bin [ndev]\n", argv[0]);
        return 2;
    }
    const int ndev = argc > 3 ? atoi(argv[3]) : 0;
    const char *rule = argc > 4 ? argv[4] : "mult";
    const int is_hals = strcmp(rule, "hals") == 0, is_pgd = strcmp(rule, "pgd") == 0, is_gram = strcmp(rule, "gram") == 0;
    if (!is_hals && !is_pgd && !is_gram && strcmp(rule, "mult") != 0) {
        fprintf(stderr, "unknown rule %s\n", rule);
        return 2;
    }
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return 3;
    int64_t dims[5];
    double reg[4];
    if (read_all(fi, dims, sizeof dims) || read_all(fi, reg, sizeof reg)) return 4;
    const int64_t N = dims[0], T = dims[1], K = dims[2], L = dims[3], iters = dims[4];
    const size_t nD = (size_t)N * T, nW = (size_t)K * N * L, nH = (size_t)K * T;
    double *data = malloc(nD * sizeof *data), *W0 = malloc(nW * sizeof *W0), *H0 = malloc(nH * sizeof *H0);
    double *W = malloc(nW * sizeof *W), *H = malloc(nH * sizeof *H);
    double *loss = malloc((size_t)(iters + 1) * sizeof *loss), *th = malloc((size_t)(iters + 1) * sizeof *th);
    if (!data || !W0 || !H0 || !W || !H || !loss || !th) return 5;
    if (read_all(fi, data, nD * 8) || read_all(fi, W0, nW * 8) || read_all(fi, H0, nH * 8)) return 4;
    fclose(fi);
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 3;

    printf("%s, %d device(s)\n", cmf_version(), cmf_device_count());
    /* build provenance and interface version, as the header promises them */
    if (cmf_abi_version() != CMF_ABI_VERSION) return 9;
    if (strstr(cmf_version(), cmf_source_digest()) == NULL || strlen(cmf_source_digest()) < 7) return 9;
    if (cmf_device_count() < 1) {
        fprintf(stderr, "no HIP device: %s\n", cmf_last_error());
        return 6;
    }
    int devices[16];
    for (int i = 0; i < 16; ++i) devices[i] = (i < cmf_device_count()) ? i : 0;
    /* a failing call must report, not crash: NULL data */
    cmf_handle bad = NULL;
    if (cmf_create(&bad, 0, N, T, K, L, NULL) != CMF_ERR_ARG || strlen(cmf_last_error()) == 0) return 7;

    for (int pass = 0; pass < 2; ++pass) {
        cmf_handle h = NULL;
        if (ndev > 0) CHECK(cmf_create_multi(&h, ndev, devices, CMF_COMM_AUTO, N, T, K, L, data));
        else CHECK(cmf_create(&h, 0, N, T, K, L, data));
        CHECK(cmf_set_factors(h, W0, H0));
        if (is_gram) CHECK(cmf_set_option(h, "gram", 1));
        if (is_hals) CHECK(cmf_set_option(h, "hals_prepare", 1)); /* the rule constructor's scratch, hals.jl:18-28 */
        if (is_pgd) CHECK(cmf_pgd_reset(h));                      /* stepW = stepH = 5, cur_loss = norm(data): pgd.jl:139-154 */
        if (is_hals || is_pgd) {
            CHECK(cmf_compute_loss(h, &loss[0]));
            for (int64_t it = 0; it < iters; ++it) {
                if (is_hals) {
                    CHECK(cmf_hals_update_motifs(h, reg[0], reg[1]));
                    CHECK(cmf_hals_update_feature_maps(h, reg[2], reg[3], &loss[it + 1]));
                } else {
                    CHECK(cmf_pgd_update_motifs(h, 1.0, 0.0, 1));
                    CHECK(cmf_pgd_update_feature_maps(h, 0.0, 0.0, 1, &loss[it + 1]));
                }
            }
            if (is_hals) {
                int64_t reruns = -1;
                CHECK(cmf_get_counter(h, "hals_pipeline_reruns", &reruns));
                if (reruns != 0) return 9;
            } else {
                double sw = 0.0, sh = 0.0;
                CHECK(cmf_pgd_get_steps(h, &sw, &sh));
                printf("steps %.17g %.17g\n", sw, sh);
            }
        } else if (pass == 0) {
            CHECK(cmf_compute_loss(h, &loss[0]));
            for (int64_t it = 0; it < iters; ++it) {
                CHECK(cmf_update_motifs(h, reg[0], reg[1]));
                CHECK(cmf_update_feature_maps(h, reg[2], reg[3], &loss[it + 1]));
            }
        } else {
            int64_t n = 0;
            int early = 0;
            CHECK(cmf_fit(h, iters, INFINITY, 0, 3, 1e-4, 0, reg[0], reg[1], reg[2], reg[3], loss, th, &n, &early));
            if (n != iters + 1 || early) return 8;
        }
        CHECK(cmf_synchronize(h));
        CHECK(cmf_get_factors(h, W, H));
        if (pass == 0) {
            char info[256];
            CHECK(cmf_comm_info(h, info, sizeof info));
            printf("%s\n", info);
        }
        CHECK(cmf_destroy(h));
        fwrite(loss, 8, (size_t)(iters + 1), fo);
        fwrite(W, 8, nW, fo);
        fwrite(H, 8, nH, fo);
    }
    fclose(fo);
    free(data); free(W0); free(H0); free(W); free(H); free(loss); free(th);
    printf("ok\n");
    return 0;
}
