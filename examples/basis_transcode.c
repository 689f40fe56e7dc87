/* Plain C99 caller of the C ABI (include/basisu_hip.h): what the body of the reference's read_to_* functions
 * (src/lib.rs:20-22, src/basis.rs:8-260) looks like on top of this library.
 *
 *   basis_transcode <rgba|etc1|etc2|uastc|astc|bc7> <in.basis> <out.bin>
 *
 * Reads a .basis file, transcodes every slice on the GPU and writes the images' bytes back to back to out.bin; prints
 * one line per image.  Exit code: 0 = done, 2 = usage / file errors, otherwise 10 + bu_status (so that a script can tell
 * "no gfx950 device" -- 17 -- from a damaged file).  There is no CPU path behind this: without a device the program says so. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "basisu_hip.h"

static int fail(bu_context* ctx, const char* what, bu_status st)
{
    fprintf(stderr, "%s: %s (status %d)%s%s\n", what, bu_status_string(st), (int)st, st == BU_ERR_HIP && ctx ? ": " : "",
            st == BU_ERR_HIP && ctx ? bu_last_error(ctx) : "");
    if (ctx) bu_context_destroy(ctx);
    return 10 + (int)st;
}

int main(int argc, char** argv)
{
    static const char* const names[6] = {"rgba", "etc1", "etc2", "uastc", "astc", "bc7"};
    bu_read_target target = BU_READ_RGBA;
    int t, found = 0;
    FILE* fp;
    long flen;
    uint8_t *file, *out;
    size_t n_images = 0, out_bytes = 0, got = 0, k;
    bu_image* images;
    bu_basis_header header;
    bu_context* ctx = NULL;
    bu_status st;

    if (argc != 4) {
        fprintf(stderr, "usage: %s <rgba|etc1|etc2|uastc|astc|bc7> <in.basis> <out.bin>\n", argv[0]);
        return 2;
    }
    for (t = 0; t < 6; t++)
        if (!strcmp(argv[1], names[t])) {
            target = (bu_read_target)t;
            found = 1;
        }
    if (!found) {
        fprintf(stderr, "unknown target %s\n", argv[1]);
        return 2;
    }
    fp = fopen(argv[2], "rb");
    if (!fp || fseek(fp, 0, SEEK_END) || (flen = ftell(fp)) < 0 || fseek(fp, 0, SEEK_SET)) {
        perror(argv[2]);
        return 2;
    }
    file = (uint8_t*)malloc(flen ? (size_t)flen : 1);
    if (!file || fread(file, 1, (size_t)flen, fp) != (size_t)flen) {
        perror(argv[2]);
        return 2;
    }
    fclose(fp);

    /* sizes first (host only: header, slice table, CRCs) ... */
    st = bu_read_query(target, file, (size_t)flen, &n_images, &out_bytes);
    if (st) return fail(NULL, "bu_read_query", st);
    /* ... then the device: no gfx950 device, no result */
    st = bu_context_create(0, &ctx);
    if (st) return fail(NULL, "bu_context_create", st);
    images = (bu_image*)calloc(n_images ? n_images : 1, sizeof *images);
    /* page-locked output: the kernels store their results straight into it (bu_host_alloc; plain malloc works too) */
    out = NULL;
    st = bu_host_alloc(ctx, out_bytes ? out_bytes : 1, (void**)&out);
    if (st || !images) return fail(ctx, "bu_host_alloc", st ? st : BU_ERR_ARGUMENT);
    st = bu_read_to(ctx, target, file, (size_t)flen, &header, images, n_images, &got, out, out_bytes);
    if (st) return fail(ctx, "bu_read_to", st);

    printf("%s: %s, %u slices, %u images -> %s, %lu bytes\n", argv[2], header.tex_format ? "UASTC" : "ETC1S", (unsigned)header.total_slices,
           (unsigned)header.total_images, names[target], (unsigned long)out_bytes);
    for (k = 0; k < got; k++)
        printf("  image %lu: %u x %u, stride %u, %lu bytes at %lu\n", (unsigned long)k, (unsigned)images[k].w, (unsigned)images[k].h,
               (unsigned)images[k].stride, (unsigned long)images[k].size, (unsigned long)images[k].offset);
    fp = fopen(argv[3], "wb");
    if (!fp || fwrite(out, 1, out_bytes, fp) != out_bytes || fclose(fp)) {
        perror(argv[3]);
        return 2;
    }
    bu_host_free(ctx, out);
    free(images);
    free(file);
    bu_context_destroy(ctx);
    return 0;
}
