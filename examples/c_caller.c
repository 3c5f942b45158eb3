/* Plain-C caller of the MI355X SuperGlue hot path: no Python, no torch.
 *
 *   gcc -std=c11 -O2 -Iinclude -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include examples/c_caller.c \
 *       -Lopenglue_amd/lib -lopenglue_amd -L/opt/rocm/lib -lamdhip64 -lm -o c_caller
 *   ./c_caller <dir> [host]
 *
 * <dir> holds raw little-endian files written by the test (tests/test_c_caller.py):
 *   shape.bin   one og_shape
 *   params.bin  fp32 arrays in state-dict order of the reference (models/superglue/superglue.py:16-23,
 *               attention_gnn.py:9-41): encoder conv weight, bias (+ BatchNorm weight, bias, running_mean,
 *               running_var for the hidden layers); per GNN layer q, k, v, out_proj (weight, bias), fc.0
 *               (weight, bias), fc.2 BatchNorm (4 arrays), fc.3 (weight, bias); linear_proj (weight, bias);
 *               mix_coefs [D] (with OG_FLAG_RESIDUAL); dustbin_score [1]
 *   inputs.bin  keypoints0, keypoints1, descriptors0, descriptors1, side_info0, side_info1, image wh (4 floats)
 * With "host" only the host-side entry points run (og_check_shape, og_packed_weights_bytes, og_pack_weights,
 * og_workspace_bytes) and packed.bin is written; otherwise the weights and inputs go to the GPU, og_forward is
 * enqueued on a stream created here, and scores.bin / matches0.bin are written.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "openglue_amd.h"

#include <hip/hip_runtime_api.h>

static void* slurp(const char* dir, const char* name, size_t* bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void* p = malloc((size_t)n + 16);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    if (bytes) *bytes = (size_t)n;
    return p;
}

static void dump(const char* dir, const char* name, const void* p, size_t bytes) {
    char path[1024];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(3); } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <dir> [host]\n", argv[0]); return 2; }
    const char* dir = argv[1];
    const int host_only = argc > 2 && !strcmp(argv[2], "host");

    size_t nb;
    og_shape* s = (og_shape*)slurp(dir, "shape.bin", &nb);
    if (nb != sizeof(og_shape)) { fprintf(stderr, "shape.bin: %zu bytes, og_shape has %zu\n", nb, sizeof(og_shape)); return 2; }
    int rc = og_check_shape(s);
    if (rc) { fprintf(stderr, "og_check_shape: %d\n", rc); return 1; }

    /* ---- host views of the parameters ---- */
    float* pf = (float*)slurp(dir, "params.bin", &nb);
    const float* cur = pf;
    const int D = s->desc_dim, L2 = 2 * s->num_stages;
    og_params P;
    memset(&P, 0, sizeof P);
    int cin = 2 + s->side_info;
    for (int i = 0; i <= s->num_hidden; ++i) {
        const int cout = i < s->num_hidden ? s->hidden[i] : D;
        P.enc_conv[i].weight = cur; cur += (size_t)cout * cin;
        P.enc_conv[i].bias = cur; cur += cout;
        if (i < s->num_hidden && !(s->flags & OG_FLAG_SIREN_ENCODER)) {
            P.enc_bn[i].weight = cur; cur += cout;
            P.enc_bn[i].bias = cur; cur += cout;
            P.enc_bn[i].running_mean = cur; cur += cout;
            P.enc_bn[i].running_var = cur; cur += cout;
        }
        cin = cout;
    }
    og_layer_params* layers = (og_layer_params*)calloc((size_t)(L2 > 0 ? L2 : 1), sizeof(og_layer_params));
    for (int l = 0; l < L2; ++l) {
        og_conv* proj[4] = {&layers[l].in_proj_q, &layers[l].in_proj_k, &layers[l].in_proj_v, &layers[l].out_proj};
        for (int p = 0; p < 4; ++p) { proj[p]->weight = cur; cur += (size_t)D * D; proj[p]->bias = cur; cur += D; }
        layers[l].fc0.weight = cur; cur += (size_t)4 * D * D; layers[l].fc0.bias = cur; cur += 2 * D;
        layers[l].fc_bn.weight = cur; cur += 2 * D; layers[l].fc_bn.bias = cur; cur += 2 * D;
        layers[l].fc_bn.running_mean = cur; cur += 2 * D; layers[l].fc_bn.running_var = cur; cur += 2 * D;
        layers[l].fc3.weight = cur; cur += (size_t)2 * D * D; layers[l].fc3.bias = cur; cur += D;
    }
    P.layers = layers;
    P.linear_proj.weight = cur; cur += (size_t)D * D; P.linear_proj.bias = cur; cur += D;
    if (s->flags & OG_FLAG_RESIDUAL) { P.mix_coefs = cur; cur += D; }
    P.dustbin_score = *cur++;
    if ((size_t)(cur - pf) * sizeof(float) != nb) { fprintf(stderr, "params.bin: consumed %zu of %zu bytes\n", (size_t)(cur - pf) * 4, nb); return 2; }

    const size_t pbytes = og_packed_weights_bytes(s), wbytes = og_workspace_bytes(s);
    if (!pbytes || !wbytes) { fprintf(stderr, "size query failed\n"); return 1; }
    void* packed_h = malloc(pbytes);
    rc = og_pack_weights(s, &P, packed_h);
    if (rc) { fprintf(stderr, "og_pack_weights: %d\n", rc); return 1; }
    printf("abi %d, packed %zu bytes, workspace %zu bytes\n", og_abi_version(), pbytes, wbytes);
    if (host_only) { dump(dir, "packed.bin", packed_h, pbytes); return 0; }

    /* ---- device side ---- */
    const size_t B = (size_t)s->batch, m = (size_t)s->m, n = (size_t)s->n, sd = (size_t)s->side_info;
    float* in_h = (float*)slurp(dir, "inputs.bin", &nb);
    const size_t sz[6] = {B * m * 2, B * n * 2, B * m * D, B * n * D, B * m * sd, B * n * sd};
    size_t tot = 4;
    for (int i = 0; i < 6; ++i) tot += sz[i];
    if (tot * sizeof(float) != nb) { fprintf(stderr, "inputs.bin: expected %zu bytes, got %zu\n", tot * 4, nb); return 2; }
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    void *packed_d, *ws, *in_d[6];
    HIP_OK(hipMalloc(&packed_d, pbytes));
    HIP_OK(hipMemcpy(packed_d, packed_h, pbytes, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&ws, wbytes));
    const float* src = in_h;
    for (int i = 0; i < 6; ++i) {
        in_d[i] = NULL;
        if (sz[i]) { HIP_OK(hipMalloc(&in_d[i], sz[i] * 4)); HIP_OK(hipMemcpy(in_d[i], src, sz[i] * 4, hipMemcpyHostToDevice)); }
        src += sz[i];
    }
    og_inputs in = {(const float*)in_d[0], (const float*)in_d[1], (const float*)in_d[2], (const float*)in_d[3],
                    (const float*)in_d[4], (const float*)in_d[5], {src[0], src[1]}, {src[2], src[3]}};
    const size_t nscores = B * (m + 1) * (n + 1);
    float *scores_d, *ms0_d;
    int64_t* m0_d;
    HIP_OK(hipMalloc((void**)&scores_d, nscores * 4));
    HIP_OK(hipMalloc((void**)&m0_d, B * m * 8));
    HIP_OK(hipMalloc((void**)&ms0_d, B * m * 4));
    og_outputs out = {scores_d, NULL, NULL, m0_d, ms0_d, NULL, NULL};
    rc = og_forward(s, &in, packed_d, ws, &out, (void*)st);          /* enqueue only */
    if (rc) { fprintf(stderr, "og_forward: %d\n", rc); return 1; }
    HIP_OK(hipStreamSynchronize(st));
    float* scores_h = (float*)malloc(nscores * 4);
    int64_t* m0_h = (int64_t*)malloc(B * m * 8);
    HIP_OK(hipMemcpy(scores_h, scores_d, nscores * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(m0_h, m0_d, B * m * 8, hipMemcpyDeviceToHost));
    dump(dir, "scores.bin", scores_h, nscores * 4);
    dump(dir, "matches0.bin", m0_h, B * m * 8);
    printf("og_forward ok: scores[0] = %g\n", scores_h[0]);
    return 0;
}
