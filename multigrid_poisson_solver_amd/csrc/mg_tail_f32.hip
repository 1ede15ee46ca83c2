// mg_tail_f32.hip -- fp32 instantiation of the coarse-tail kernel (source: mg_tail_impl.h), the
// coarse part of the mixed-precision mode's fp32 cycle
#define MG_REAL float
#define MG_REAL_NS f32
#include "mg_tail_impl.h"

namespace mg {
namespace k {

void tail_launch_f32(hipStream_t s, const TailArgsF &a) { f32::tail_launch(s, a); }

}  // namespace k
}  // namespace mg
