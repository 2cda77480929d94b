// kernels_quant.hip — k-means, PQ lookup-table and ADC kernels (filled in below).
#include "kernels.hpp"
namespace comet {}
