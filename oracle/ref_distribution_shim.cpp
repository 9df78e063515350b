// ref_distribution_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" doorway onto the reference's own header-only index math
// (include/dlaf/matrix/util_distribution.h:29-196), compiled from the header where it
// lies under /root/reference (oracle/Makefile target `ref`).  Used by tests/test_oracle.py
// in the build container to validate the C restatement in dlaf_oracle.c; no reference
// source is copied into this repository.
#include <dlaf/matrix/util_distribution.h>

namespace u = dlaf::util::matrix;

extern "C" {
long ref_tile_from_element(long e, long ts, long off) { return u::tile_from_element(e, ts, off); }
long ref_tile_element_from_element(long e, long ts, long off) { return u::tile_element_from_element(e, ts, off); }
long ref_element_from_tile_and_tile_element(long t, long te, long ts, long off) {
  return u::element_from_tile_and_tile_element(t, te, ts, off);
}
int ref_rank_global_tile(long gt, long tpb, int gs, int src, long toff) {
  return u::rank_global_tile(gt, tpb, gs, src, toff);
}
long ref_local_tile_from_global_tile(long gt, long tpb, int gs, int rank, int src, long toff) {
  return u::local_tile_from_global_tile(gt, tpb, gs, rank, src, toff);
}
long ref_next_local_tile_from_global_tile(long gt, long tpb, int gs, int rank, int src, long toff) {
  return u::next_local_tile_from_global_tile(gt, tpb, gs, rank, src, toff);
}
long ref_global_tile_from_local_tile(long lt, long tpb, int gs, int rank, int src, long toff) {
  return u::global_tile_from_local_tile(lt, tpb, gs, rank, src, toff);
}
}
