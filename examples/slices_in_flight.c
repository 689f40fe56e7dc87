/* Plain C99 caller of the C ABI (include/basisu_hip.h): the reference's loop over the slices of a texture array
 * (src/basis.rs:246-257: one transcode per slice, one after another) with the slices resident on the GPU and FOUR of them in
 * flight -- each slice is one launch, slice i goes to the context's stream i % 4, under the shared launch policy.
 *
 *   slices_in_flight <astc|bc7|etc1|etc2> <in.uastc> <n_slices> <out.bin> [batch]
 *
 * With `batch` as a fifth argument the loop is ONE call of bu_uastc_transcode_batch_in_flight (the library merges, groups and cuts the slices and issues
 * the launches on its four streams itself); without it the program issues one launch per slice on the streams it got from bu_context_stream.
 *
 * in.uastc holds n_slices equal slices of raw UASTC blocks (16 bytes each) back to back; out.bin receives the transcoded
 * slices back to back.  Exit code: 0 = done, 2 = usage / file errors, otherwise 10 + bu_status.  No HIP header is needed:
 * device memory, streams and the final wait all come from the library (bu_device_alloc, bu_context_stream,
 * bu_context_synchronize).  For the launches to overlap the process must own one hardware queue per stream: run it with
 * GPU_MAX_HW_QUEUES=8 in the environment. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "basisu_hip.h"

#define STREAMS 4

static int fail(bu_context* ctx, const char* what, bu_status st)
{
    fprintf(stderr, "%s: %s (status %d)%s%s\n", what, bu_status_string(st), (int)st, st == BU_ERR_HIP && ctx ? ": " : "",
            st == BU_ERR_HIP && ctx ? bu_last_error(ctx) : "");
    if (ctx) bu_context_destroy(ctx);
    return 10 + (int)st;
}

int main(int argc, char** argv)
{
    static const char* const names[4] = {"astc", "bc7", "etc1", "etc2"};
    bu_target target = BU_TARGET_BC7;
    int t, found = 0;
    long flen, n_slices;
    size_t slice_bytes, n_blocks, out_slice_bytes, i;
    uint8_t *in, *out;
    void *d_in = NULL, *d_out = NULL, *d_status = NULL, *streams[STREAMS];
    uint64_t word = 0, first_bad = 0;
    bu_context* ctx = NULL;
    bu_status st;
    FILE* fp;

    if (argc != 5 && !(argc == 6 && !strcmp(argv[5], "batch"))) {
        fprintf(stderr, "usage: %s <astc|bc7|etc1|etc2> <in.uastc> <n_slices> <out.bin> [batch]\n", argv[0]);
        return 2;
    }
    for (t = 0; t < 4; t++)
        if (!strcmp(argv[1], names[t])) {
            target = (bu_target)t;
            found = 1;
        }
    n_slices = strtol(argv[3], NULL, 10);
    if (!found || n_slices < 1) {
        fprintf(stderr, "unknown target or slice count\n");
        return 2;
    }
    fp = fopen(argv[2], "rb");
    if (!fp || fseek(fp, 0, SEEK_END) || (flen = ftell(fp)) < 0 || fseek(fp, 0, SEEK_SET)) {
        perror(argv[2]);
        return 2;
    }
    if (flen == 0 || flen % (16 * n_slices) != 0) {
        fprintf(stderr, "%s: %ld bytes are not %ld equal slices of 16-byte blocks\n", argv[2], flen, n_slices);
        return 2;
    }
    slice_bytes = (size_t)flen / (size_t)n_slices;
    n_blocks = slice_bytes / 16;
    out_slice_bytes = n_blocks * bu_target_block_bytes(target);
    in = (uint8_t*)malloc((size_t)flen);
    out = (uint8_t*)malloc(out_slice_bytes * (size_t)n_slices);
    if (!in || !out || fread(in, 1, (size_t)flen, fp) != (size_t)flen) {
        perror(argv[2]);
        return 2;
    }
    fclose(fp);

    st = bu_context_create(0, &ctx);
    if (st) return fail(NULL, "bu_context_create", st);
    /* the launch policy stays at its default, BU_LAUNCH_AUTO: a launch on one of the context's streams takes the half-CU shape exactly when
       another of them has work in flight (here: every launch but the first) */
    for (t = 0; t < STREAMS; t++) {
        st = bu_context_stream(ctx, t, &streams[t]);
        if (st) return fail(ctx, "bu_context_stream", st);
    }
    {   /* does every stream have a hardware queue of its own?  The context saw to that when it created them (ordinary streams if the runtime's
           pool has room, CU-mask streams otherwise); results never depend on it, the overlap does */
        int effective = 0, mode = 0;
        st = bu_context_query_in_flight(ctx, STREAMS, &effective, &mode);
        if (st) return fail(ctx, "bu_context_query_in_flight", st);
        if (effective < STREAMS)
            fprintf(stderr, "note: the %d streams keep only %d launches in flight in this process (%s streams)\n", STREAMS, effective,
                    mode == BU_STREAM_QUEUE_CU_MASK ? "CU-mask" : "ordinary");
    }
    st = bu_device_alloc(ctx, (size_t)flen, &d_in);
    if (!st) st = bu_device_alloc(ctx, out_slice_bytes * (size_t)n_slices, &d_out);
    if (!st) st = bu_device_alloc(ctx, sizeof(uint64_t), &d_status);
    if (st) return fail(ctx, "bu_device_alloc", st);
    st = bu_memcpy(ctx, d_in, in, (size_t)flen, 1);
    if (st) return fail(ctx, "bu_memcpy", st);
    /* one status word for the whole array: slice i reports block indices from i * n_blocks on */
    st = bu_status_word_reset(ctx, (uint64_t*)d_status, streams[0]);
    if (!st) st = bu_context_synchronize(ctx); /* the other streams must see the reset */
    if (st) return fail(ctx, "bu_status_word_reset", st);
    if (argc == 6) { /* the whole loop in one call: slice table in, launches on the context's streams out */
        const void** ins = (const void**)malloc(sizeof(void*) * (size_t)n_slices);
        void** outs = (void**)malloc(sizeof(void*) * (size_t)n_slices);
        size_t* counts = (size_t*)malloc(sizeof(size_t) * (size_t)n_slices);
        if (!ins || !outs || !counts) return fail(ctx, "malloc", BU_ERR_ARGUMENT);
        for (i = 0; i < (size_t)n_slices; i++) {
            ins[i] = (const uint8_t*)d_in + i * slice_bytes;
            outs[i] = (uint8_t*)d_out + i * out_slice_bytes;
            counts[i] = n_blocks;
        }
        st = bu_uastc_transcode_batch_in_flight(ctx, target, (size_t)n_slices, ins, counts, outs, 0, NULL, (uint64_t*)d_status, STREAMS);
        free(ins);
        free(outs);
        free(counts);
        if (st) return fail(ctx, "bu_uastc_transcode_batch_in_flight", st);
    } else {
        for (i = 0; i < (size_t)n_slices; i++) {
            st = bu_uastc_transcode_device(ctx, target, (const uint8_t*)d_in + i * slice_bytes, n_blocks, (uint8_t*)d_out + i * out_slice_bytes, 0,
                                           (uint64_t)i * n_blocks, (uint64_t*)d_status, streams[i % STREAMS]);
            if (st) return fail(ctx, "bu_uastc_transcode_device", st);
        }
    }
    st = bu_context_synchronize(ctx);
    if (st) return fail(ctx, "bu_context_synchronize", st);
    st = bu_memcpy(ctx, &word, d_status, sizeof(word), 0);
    if (!st) st = bu_status_word_decode(word, &first_bad);
    if (st) {
        fprintf(stderr, "block %llu of the array (slice %llu): ", (unsigned long long)first_bad, (unsigned long long)(first_bad / n_blocks));
        return fail(ctx, "transcode", st); /* uastc.rs:157-165: the first failing block decides */
    }
    st = bu_memcpy(ctx, out, d_out, out_slice_bytes * (size_t)n_slices, 0);
    if (st) return fail(ctx, "bu_memcpy", st);
    bu_device_free(ctx, d_in);
    bu_device_free(ctx, d_out);
    bu_device_free(ctx, d_status);
    bu_context_destroy(ctx);

    fp = fopen(argv[4], "wb");
    if (!fp || fwrite(out, 1, out_slice_bytes * (size_t)n_slices, fp) != out_slice_bytes * (size_t)n_slices || fclose(fp)) {
        perror(argv[4]);
        return 2;
    }
    printf("%ld slices x %lu blocks -> %s, %d launches in flight\n", n_slices, (unsigned long)n_blocks, argv[1], STREAMS);
    free(in);
    free(out);
    return 0;
}
