"""2-D block-cyclic index helpers, served by the C++ host (csrc/host/distribution.hpp) through the
C ABI -- reference: include/dlaf/matrix/util_distribution.h:82-196 (tiles_per_block = 1, no offset)."""
from .capi import lib


def rank_global_tile(global_tile: int, grid_size: int, src_rank: int = 0) -> int:
    return lib().dlaf_mi355x_dist_owner(global_tile, grid_size, src_rank)


def local_tile_from_global_tile(global_tile: int, grid_size: int, rank: int, src_rank: int = 0) -> int:
    return lib().dlaf_mi355x_dist_local_tile(global_tile, grid_size, rank, src_rank)


def next_local_tile_from_global_tile(global_tile: int, grid_size: int, rank: int, src_rank: int = 0) -> int:
    return lib().dlaf_mi355x_dist_next_local_tile(global_tile, grid_size, rank, src_rank)


def global_tile_from_local_tile(local_tile: int, grid_size: int, rank: int, src_rank: int = 0) -> int:
    return lib().dlaf_mi355x_dist_global_tile(local_tile, grid_size, rank, src_rank)


def local_size(n: int, nb: int, grid_size: int, rank: int, src_rank: int = 0) -> int:
    return lib().dlaf_mi355x_dist_local_size(n, nb, grid_size, rank, src_rank)


def local_nr_tiles(n: int, nb: int, grid_size: int, rank: int, src_rank: int = 0) -> int:
    return lib().dlaf_mi355x_dist_local_tiles(n, nb, grid_size, rank, src_rank)
