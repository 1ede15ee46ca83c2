// mg_tail.hip -- fp64 instantiation of the coarse-tail kernel (source: mg_tail_impl.h)
#define MG_REAL double
#define MG_REAL_NS f64
#include "mg_tail_impl.h"

namespace mg {
namespace k {

bool tail_fits(const TailArgs &a) { return f64::tail_lds_bytes(a) <= (size_t)(160 * 1024 - 512); }
void tail_launch(hipStream_t s, const TailArgs &a) { f64::tail_launch(s, a); }

}  // namespace k
}  // namespace mg
