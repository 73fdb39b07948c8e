// bfpq_fused_dt.hip -- the fused single-pass kernels of ONE dtype (compile with -DBFPQ_FUSED_DT=0|1|2: f32, f16, bf16).
// Three objects instead of one translation unit with every instantiation: they build side by side.
#include <hip/hip_runtime.h>
#include "bfpq.h"
#include "bfpq_common.h"
#include "bfpq_fused.h"

#ifndef BFPQ_FUSED_DT
#error "compile with -DBFPQ_FUSED_DT=0, 1 or 2"
#endif
#define BFPQ_CAT_(a, b) a##b
#define BFPQ_CAT(a, b) BFPQ_CAT_(a, b)

namespace bfpq_dev {

int BFPQ_CAT(fused_launch_, BFPQ_FUSED_DT)(const FusedArgs& a, int M, bool sfirst, hipStream_t s) { return launch_fused<BFPQ_FUSED_DT>(a, M, sfirst, s); }
int BFPQ_CAT(fused_threshold_, BFPQ_FUSED_DT)(const FusedArgs& a, hipStream_t s) { return launch_fused_threshold<BFPQ_FUSED_DT>(a, s); }
int BFPQ_CAT(fused_mx8_, BFPQ_FUSED_DT)(const FusedArgs& a, hipStream_t s) { return launch_fused_mx8<BFPQ_FUSED_DT>(a, s); }
int BFPQ_CAT(fused_batched_, BFPQ_FUSED_DT)(const FusedArgs& a, const BatchArgs& b, int M, bool sfirst, hipStream_t s)
{
    return launch_batched_dt<BFPQ_FUSED_DT>(a, b, M, sfirst, s);
}

}  // namespace bfpq_dev
