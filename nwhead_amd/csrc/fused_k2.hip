// fused_k2.hip -- nw_fused_kernel instantiations for NW_SCORE_COSINE (gfx950 / MI355X only).
#include "fused_impl.h"
namespace nw {
NW_INSTANTIATE_FUSED_KIND(NW_SCORE_COSINE)
}
