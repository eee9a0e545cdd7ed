// compile_check.hip -- keeps the retired kernels building against the product headers:
//   hipcc --offload-arch=gfx950 -std=c++17 -ffp-contract=off -c -o /dev/null tools/experiments/retired/compile_check.hip
#include "ell_retired_kernels.hpp"
#include "ellstable_retired_kernels.hpp"
using namespace ellhip;
template __global__ void ellhip::k_symv_tail<2, true, 16>(const double*, long long, long long, const double*, double*, double*, double*,
                                                          const double*, double*, DevState*, SymvTailCtl*, unsigned, unsigned);
template __global__ void ellhip::k_symv_reduce_scalar<16>(long long, long long, const double*, const double*, double*, DevState*,
                                                          const double*, double*, double*, double*, double*, EllCalcDev,
                                                          const CutParams*, CutParams, int, int, int*, double*, unsigned*, unsigned);
int main() { return 0; }
