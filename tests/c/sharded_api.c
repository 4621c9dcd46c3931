/*
 * The sharded register driven from plain C through include/qcx.h only: the same circuit (qcx_quantum_computation, i.e.
 * the schedule of qc_shor.c:712-737) and the same measurement stream on an unsharded register and on registers sharded
 * 2, 4 and 8 ways (spread over the visible GPUs: qcx_spread_devices; one GPU = all on device 0), with and without relay
 * striping (relays = GPUs that hold no shard, or device 0 again when there is none).  Amplitudes and measured indices must
 * be identical, bit for bit.  Prints "ok" and exits 0, or says what differed.  tests/test_gpu_sharded_c.py runs it.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qcx.h"

#define CHECK(call) do { int s_ = (call); if (s_ != QCX_NO_ERROR) { \
    fprintf(stderr, "%s -> %s (%s)\n", #call, qcx_status_string(s_), qcx_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 11, M = argc > 2 ? atoi(argv[2]) : 5;
    const unsigned C = 21, a = 2;
    const unsigned long dim = 1ul << (L + M);
    double *ref = malloc(2 * dim * sizeof(double)), *got = malloc(2 * dim * sizeof(double));
    if (!ref || !got) return 1;

    qcx_register *plain = NULL;
    qcx_rng *rng = qcx_rng_alloc();
    unsigned long want_idx[3], idx;
    CHECK(qcx_register_create(L, M, &plain));
    CHECK(qcx_reset_register(plain));
    CHECK(qcx_quantum_computation(C, a, 0, plain));
    CHECK(qcx_state_read(plain, 0, dim, ref));
    qcx_rng_set(rng, 4242);
    for (int shot = 0; shot < 3; shot++) {
        CHECK(qcx_reset_register(plain));
        CHECK(qcx_quantum_computation(C, a, 0, plain));
        CHECK(qcx_measure_state(plain, rng, &want_idx[shot]));
    }
    CHECK(qcx_register_destroy(plain));

    int visible = 0;
    CHECK(qcx_device_count(&visible));
    for (unsigned shards = 2; shards <= 8; shards *= 2) {
        for (int relays = 0; relays <= 1; relays++) {
            qcx_register *reg = NULL;
            int devs[16];
            CHECK(qcx_spread_devices(shards, visible, devs));
            CHECK(qcx_register_create_sharded(L, M, shards, relays ? devs : NULL, &reg));      /* NULL = the same spreading */
            if (qcx_register_shards(reg) != shards) { fprintf(stderr, "shards\n"); return 1; }
            if (relays) {                        /* two relays: the highest-numbered GPUs that hold no shard, else device 0 */
                int rd[2] = {0, 0}, used = devs[shards - 1] + 1;
                if (visible - used >= 1) rd[0] = visible - 1;
                if (visible - used >= 2) rd[1] = visible - 2; else rd[1] = rd[0];
                CHECK(qcx_sharded_set_relays(reg, 2, rd));
            }
            CHECK(qcx_reset_register(reg));
            CHECK(qcx_quantum_computation(C, a, 0, reg));
            CHECK(qcx_state_read(reg, 0, dim, got));
            if (memcmp(ref, got, 2 * dim * sizeof(double)) != 0) { fprintf(stderr, "amplitudes differ at %u shards (relays %d)\n", shards, relays); return 1; }
            qcx_rng_set(rng, 4242);
            for (int shot = 0; shot < 3; shot++) {
                CHECK(qcx_reset_register(reg));
                CHECK(qcx_quantum_computation(C, a, 0, reg));
                CHECK(qcx_measure_state(reg, rng, &idx));
                if (idx != want_idx[shot]) { fprintf(stderr, "shot %d: %lu vs %lu at %u shards\n", shot, idx, want_idx[shot], shards); return 1; }
            }
            unsigned long ex = 0, packs = 0;
            CHECK(qcx_sharded_stats(reg, &ex, &packs));
            if (ex == 0) { fprintf(stderr, "no exchange happened\n"); return 1; }
            double p = 0.0;
            CHECK(qcx_total_probability(reg, &p));          /* collapsed state: exactly 1 */
            if (p != 1.0) { fprintf(stderr, "total probability %.17g\n", p); return 1; }
            CHECK(qcx_register_destroy(reg));
        }
    }
    qcx_rng_free(rng);
    free(ref); free(got);
    printf("ok\n");
    return 0;
}
