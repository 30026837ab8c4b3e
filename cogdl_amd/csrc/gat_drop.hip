// gat_drop.hip -- the attention-dropout instantiations of the fused GAT functors (gat_op.h, DROP = true): the branch
// CogDL's gat model takes by default (attn_drop 0.5: cogdl/models/nn/gat.py:30, cogdl/layers/gat_layer.py:72-77), as ONE
// forward kernel and two backward passes with the mask regenerated from the seed (philox.h) instead of the
// [E,H] score / attention / mask tensors and torch's indexing backward.
#include "gat_op.h"

namespace cogdl {

int gat_fwd_drop(const GatFwdArgs &a, int dtype, void *ws, size_t wsb, hipStream_t s) {
    return gat_fwd_any<true>(a, dtype, ws, wsb, s);
}

int gat_bwd_drop(const GatBwdArgs &b, const GatBwdGeometry &g, int dtype, hipStream_t s) {
    return gat_bwd_any<true>(b, g, dtype, s);
}

}  // namespace cogdl
