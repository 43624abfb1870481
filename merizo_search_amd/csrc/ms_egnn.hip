// placeholder until the encoder kernels land (next commit): keeps the ABI complete.
#include "ms_common.h"
extern "C" {
size_t ms_egnn_weight_floats(void) { return 2 * 396165; }
size_t ms_egnn_prepared_bytes(void) { return 16; }
int ms_egnn_prepare_weights(const float *, void *, ms_stream_t) { MS_FAIL(MS_ERR_ARG, "ms_egnn: not built yet"); }
size_t ms_egnn_workspace_bytes(int, int64_t, int64_t) { return 16; }
int ms_egnn_embed(const void *, const float *, int, const float *, const int32_t *, const int32_t *, int, float *,
                  void *, size_t, ms_stream_t) { MS_FAIL(MS_ERR_ARG, "ms_egnn: not built yet"); }
}
