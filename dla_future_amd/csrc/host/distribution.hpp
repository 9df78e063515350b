// distribution.hpp -- 2-D block-cyclic index math for one axis of a distributed matrix.
//
// Re-states what the Cholesky path needs of the reference's Distribution
// (include/dlaf/matrix/util_distribution.h:82-196, src/matrix/distribution.cpp:109-151) for the
// only configuration that path accepts (cholesky.h:39-79: square matrix, square blocks,
// block == tile, no sub-matrix offset): tile g of an axis lives on process
// (g + src) mod P at local index g / P.  Verified against the reference's golden rows by
// tests/test_host_logic.py through the dlaf_mi355x_dist_* C entry points.
#pragma once
#include <algorithm>
#include <cstdint>

namespace dlaf_mi355x {

struct Axis {
  long n = 0;    // elements along the axis
  int nb = 1;    // tile size
  int P = 1;     // processes along the axis
  int rank = 0;  // this process
  int src = 0;   // process owning tile 0

  long nt() const { return n > 0 ? (n + nb - 1) / nb : 0; }
  // position of this process counted from the source process
  int shift() const { return (rank + P - src) % P; }
  int owner(long g) const { return (int) ((g + src) % P); }
  bool mine(long g) const { return owner(g) == rank; }
  // local index of global tile g, -1 when owned elsewhere (local_tile_from_global_tile)
  long local_of(long g) const { return mine(g) ? g / P : -1; }
  // local index of the first tile with global index >= g (next_local_tile_from_global_tile)
  long next_local(long g) const {
    const long cycle = g / P;
    const int pos = (int) (g % P);
    return shift() >= pos ? cycle : cycle + 1;
  }
  // global index of local tile l (global_tile_from_local_tile)
  long global_of(long l) const { return l * P + shift(); }
  long local_tiles() const { return next_local(nt()); }
  int tile_extent(long g) const { return (int) std::min<long>(nb, n - g * nb); }
  int last_extent() const { return nt() > 0 ? tile_extent(nt() - 1) : 0; }
  // local number of elements (compute_local_size)
  long local_size() const {
    const long lt = local_tiles();
    if (lt == 0)
      return 0;
    long sz = lt * nb;
    if (mine(nt() - 1))
      sz -= nt() * nb - n;
    return sz;
  }
};

}  // namespace dlaf_mi355x
