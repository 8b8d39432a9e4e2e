/* examples/eval_packed.c — the C ABI of include/kz_hip.h from plain C (what any FFI sees): load a model, create an
 * engine, evaluate a batch of packed boards, read the results in place, clean up.  Every call returns 0 or sets
 * kz_last_error().
 *
 *   gcc -std=c99 -I include examples/eval_packed.c -L kzero_amd -lkzhip -Wl,-rpath,$PWD/kzero_amd -o /tmp/eval_packed
 *   /tmp/eval_packed tests/golden/ataxx7_4x64.kzm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kz_hip.h"

#define CHECK(call)                                                      \
    do {                                                                 \
        if ((call) != 0) {                                               \
            fprintf(stderr, "%s failed: %s\n", #call, kz_last_error()); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s model.kzm|model.onnx [input_scalar_channels for ONNX]\n", argv[0]);
        return 2;
    }
    kz_model *model = NULL;
    const size_t n = strlen(argv[1]);
    if (n > 5 && strcmp(argv[1] + n - 5, ".onnx") == 0) CHECK(kz_model_load_onnx(argv[1], argc > 2 ? atoi(argv[2]) : 0, &model));
    else CHECK(kz_model_load(argv[1], &model));
    kz_model_info info;
    CHECK(kz_model_get_info(model, &info));
    printf("model: %d planes (%d scalar + %d bool) on %dx%d, tower %dx%d, policy %d\n", info.input_channels,
           info.input_scalar_channels, info.input_bool_channels, info.board_h, info.board_w, info.tower_depth,
           info.tower_channels, info.policy_len);

    const int batch = 4;
    kz_engine *engine = NULL;
    CHECK(kz_engine_create(model, 0, batch, KZ_DTYPE_F32, &engine));

    /* packed boards: BitBuffer storage (bit i of the bool planes = bit i%8 of byte i/8) + the scalar planes' values */
    unsigned char *bits = calloc((size_t)batch, (size_t)info.bits_bytes);
    float *scalars_in = calloc((size_t)batch * (size_t)(info.input_scalar_channels ? info.input_scalar_channels : 1), sizeof(float));
    for (int b = 0; b < batch; b++) bits[(size_t)b * info.bits_bytes] = (unsigned char)(1u << b); /* one piece each */

    /* asynchronous pair + zero-copy view; kz_engine_eval_packed is the one-call form with caller buffers */
    CHECK(kz_engine_submit_packed(engine, 0, bits, (size_t)info.bits_bytes, scalars_in, batch));
    const float *scalars_out = NULL, *policy = NULL;
    CHECK(kz_engine_wait_view(engine, 0, &scalars_out, &policy));
    for (int b = 0; b < batch; b++)
        printf("board %d: value logit %+.4f  wdl logits %+.4f %+.4f %+.4f  moves left %+.4f  policy[0] %+.4f\n", b,
               scalars_out[b * 5], scalars_out[b * 5 + 1], scalars_out[b * 5 + 2], scalars_out[b * 5 + 3],
               scalars_out[b * 5 + 4], policy[(size_t)b * info.policy_len]);

    free(bits);
    free(scalars_in);
    kz_engine_destroy(engine);
    kz_model_free(model);
    return 0;
}
