/* A plain-C client of libgypsum_hip: the boundary of include/gypsum_hip.h used without Python.
 *
 *   gcc -std=c11 -Iinclude examples/c_client.c -o c_client -Lgypsum_amd/csrc -lgypsum_hip -Wl,-rpath,$PWD/gypsum_amd/csrc -lm
 *   ./c_client
 *
 * Generates one 2.046 Msps stream with four satellites into HBM, runs the full 32-satellite acquisition search
 * (acquisition.py:52-152), tracks the four strongest for 300 ms with the device-resident loops (tracker.py:331-389),
 * feeds the per-millisecond records to the native navigation-bit integrator, and prints one line per stage.
 * Exit status 0 = the planted satellites were found at their Doppler bin / code phase and are still tracked. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gypsum_hip.h"

#define CHECK(call)                                                                    \
    do {                                                                               \
        int rc_ = (call);                                                              \
        if (rc_ != GYP_OK) {                                                           \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, gyp_last_error(ctx)); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

int main(void) {
    gyp_ctx* ctx = NULL;
    const int64_t fs = 2046000;
    const int32_t n = 2046, n_ms = 310;
    if (gyp_create(0, &ctx) != GYP_OK) {
        fprintf(stderr, "gyp_create: %s\n", gyp_last_error(NULL));
        return 2;
    }
    CHECK(gyp_set_stream_format(ctx, fs, n));

    gyp_synth_sat sats[4] = {
        {5, 100, -2500.0, 0.3, 0.010f, 3}, {12, 1500, 1500.0, 2.0, 0.010f, 11}, {23, 2045, 4000.0, -1.0, 0.010f, 0}, {30, 0, 0.0, 0.5, 0.010f, 19}};
    void* iq = NULL;
    CHECK(gyp_malloc(ctx, (uint64_t)n_ms * n * 8, &iq));
    CHECK(gyp_synth_iq_dev(ctx, (float*)iq, 1, (int64_t)n_ms * n, n_ms, sats, 4, 0.02f, 20260925ull));

    int32_t ids[32];
    for (int i = 0; i < 32; ++i) ids[i] = i + 1;
    void* acq_dev = NULL;
    CHECK(gyp_malloc(ctx, 32 * sizeof(gyp_acq_result), &acq_dev));
    CHECK(gyp_acquire_dev(ctx, (const float*)iq, 1, (int64_t)n_ms * n, 10, ids, 32, (gyp_acq_result*)acq_dev));
    gyp_acq_result acq[32];
    CHECK(gyp_memcpy_d2h(ctx, acq, acq_dev, sizeof acq));

    int bad = 0;
    gyp_chan_init inits[4];
    for (int k = 0; k < 4; ++k) {
        const gyp_acq_result* a = &acq[sats[k].sat_id - 1];
        printf("acquired sv%02d: doppler %d Hz, code phase %d, strength %.2f\n", a->sat_id, a->doppler_hz, a->code_phase, a->strength);
        if (fabs(a->doppler_hz - sats[k].doppler_hz) > 60.0 || a->code_phase != sats[k].code_phase || a->strength < 3.0) ++bad;
        inits[k].stream = 0; inits[k].sat_id = a->sat_id; inits[k].doppler_hz = a->doppler_hz;
        inits[k].carrier_phase = a->carrier_phase; inits[k].code_phase = a->code_phase; inits[k].reserved = 0;
    }

    gyp_bank* bank = NULL;
    CHECK(gyp_bank_create(ctx, inits, 4, &bank));
    const int32_t t_ms = n_ms - 10;
    double* t0 = malloc(sizeof(double) * t_ms);
    double* t1 = malloc(sizeof(double) * t_ms);
    for (int i = 0; i < t_ms; ++i) {
        t0[i] = (double)(10 + i) * n / (double)fs;
        t1[i] = (double)(11 + i) * n / (double)fs;
    }
    void *t_dev = NULL, *rec_dev = NULL;
    CHECK(gyp_malloc(ctx, sizeof(double) * t_ms, &t_dev));
    CHECK(gyp_malloc(ctx, sizeof(gyp_track_rec) * 4 * t_ms, &rec_dev));
    CHECK(gyp_memcpy_h2d(ctx, t_dev, t0, sizeof(double) * t_ms));
    CHECK(gyp_track_block_dev(bank, (const float*)iq + (size_t)10 * n * 2, (int64_t)n_ms * n, t_ms, (const double*)t_dev, (gyp_track_rec*)rec_dev));
    gyp_track_rec* rec = malloc(sizeof(gyp_track_rec) * 4 * t_ms);
    CHECK(gyp_memcpy_d2h(ctx, rec, rec_dev, sizeof(gyp_track_rec) * 4 * t_ms));

    gyp_bits* bits = NULL;
    if (gyp_bits_create(4, &bits) != GYP_OK) return 2;
    gyp_bit_event ev[64];
    int32_t n_ev = 0;
    if (gyp_bits_push_block(bits, rec, 4, t_ms, t0, t1, ev, 64, &n_ev) != GYP_OK) {
        fprintf(stderr, "gyp_bits_push_block: %s\n", gyp_last_error(NULL));
        return 2;
    }
    for (int k = 0; k < 4; ++k) {
        const gyp_track_rec* last = &rec[(size_t)k * t_ms + t_ms - 1];
        int flips = 0;
        for (int i = 101; i < t_ms; ++i)   /* after pull-in, symbols change only at 20 ms bit edges */
            flips += rec[(size_t)k * t_ms + i].pseudosymbol != rec[(size_t)k * t_ms + i - 1].pseudosymbol;
        printf("tracked  sv%02d: doppler %.2f Hz, locked %d, status %d, %d symbol transitions in %d ms\n", inits[k].sat_id,
               last->doppler_hz, last->locked, last->status, flips, t_ms - 101);
        if (last->status != 0 || fabs(last->doppler_hz - sats[k].doppler_hz) > 40.0) ++bad;   /* the Costas loop is still pulling in */
    }
    printf("navigation bits emitted: %d\n", n_ev);
    gyp_bits_destroy(bits);
    gyp_bank_destroy(bank);
    gyp_free(ctx, iq); gyp_free(ctx, acq_dev); gyp_free(ctx, t_dev); gyp_free(ctx, rec_dev);
    free(t0); free(t1); free(rec);
    gyp_destroy(ctx);
    printf(bad ? "FAILED (%d checks)\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
