"""Row tiling of a frame across ranks (one process per GPU) and the gather of the tiles on rank 0.

The reference parallelises over image rows with OpenMP (reference src/render.cpp:86-89); across GPUs the
same independence is used: the scene is replicated, the frame is cut into blocks of ROW_BLOCK consecutive
Screen rows, block b belongs to rank b % world (interleaved, so every rank gets a share of the busy middle of
the picture), and the only exchange is one gather of the rendered blocks to rank 0.  Random streams are keyed
by absolute pixel, so the assembled frame does not depend on the number of ranks.
"""
import torch
import torch.distributed as dist

ROW_BLOCK = 16


def num_blocks(height, block=ROW_BLOCK):
    return (height + block - 1) // block


def blocks_of_rank(rank, world, height, block=ROW_BLOCK):
    return list(range(rank, num_blocks(height, block), world))


def blocks_per_rank(world, height, block=ROW_BLOCK):
    return (num_blocks(height, block) + world - 1) // world


def block_rows(b, height, block=ROW_BLOCK):
    """Block b as (screen_top, screen_bottom, row_begin, row_end): Screen rows [top, bottom) counted from the
    top of the picture, and the same rows counted from the bottom, which is what rt_render takes."""
    top, bottom = b * block, min(height, (b + 1) * block)
    return top, bottom, height - bottom, height - top


def local_buffer(world, width, height, device, block=ROW_BLOCK):
    """Per-rank tile store: blocks_per_rank slots of `block` rows each (the last block may be short)."""
    return torch.zeros((blocks_per_rank(world, height, block) * block, width, 3), dtype=torch.float32, device=device)


def render_local(render_rows, rank, world, width, height, local, block=ROW_BLOCK):
    """Calls render_rows(row_begin, row_end, out) for every block of this rank; `out` is the [rows, width, 3]
    view of `local` the block's pixels go to (top-most row first)."""
    results = []
    for k, b in enumerate(blocks_of_rank(rank, world, height, block)):
        top, bottom, row_begin, row_end = block_rows(b, height, block)
        results.append(render_rows(row_begin, row_end, local[k * block:k * block + (bottom - top)]))
    return results


def gather_buffers(local, world):
    """Receive buffers for gather_frame that can be kept across frames: one tensor holding every rank's tile store
    (so that the frame can be put together with a single row gather) and the per-rank views torch.distributed
    wants."""
    store = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    return store, [store[r] for r in range(world)]


def frame_row_index(world, height, device, block=ROW_BLOCK):
    """For every Screen row of the frame, its row in the stacked tile stores [world * rows_per_rank]."""
    per_rank = blocks_per_rank(world, height, block) * block
    idx = torch.empty(height, dtype=torch.long)
    for r in range(world):
        for k, b in enumerate(blocks_of_rank(r, world, height, block)):
            top, bottom, _, _ = block_rows(b, height, block)
            idx[top:bottom] = torch.arange(r * per_rank + k * block, r * per_rank + k * block + (bottom - top))
    return idx.to(device)


def gather_frame(local, rank, world, width, height, block=ROW_BLOCK, gathered=None, row_index=None):
    """Gathers every rank's tile store on rank 0 and returns the assembled [height, width, 3] frame there
    (None elsewhere).  `gathered` (from gather_buffers) and `row_index` (from frame_row_index) may be kept across
    frames; with both, putting the frame together is one row gather on the device."""
    if world == 1:
        return local[:height]
    if rank == 0 and gathered is None:
        gathered = gather_buffers(local, world)
    dist.gather(local, gathered[1] if rank == 0 else None, dst=0)
    if rank != 0:
        return None
    if row_index is None:
        row_index = frame_row_index(world, height, local.device, block)
    store = gathered[0]
    return store.view(world * store.shape[1], width, 3).index_select(0, row_index)


def assemble(gathered, world, width, height, block=ROW_BLOCK):
    """Reference form of the assembly, block by block (tests compare gather_frame against it)."""
    frame = torch.empty((height, width, 3), dtype=gathered[0].dtype, device=gathered[0].device)
    for r in range(world):
        for k, b in enumerate(blocks_of_rank(r, world, height, block)):
            top, bottom, _, _ = block_rows(b, height, block)
            frame[top:bottom] = gathered[r][k * block:k * block + (bottom - top)]
    return frame


def native_comm(lib, rank, world):
    """A communicator of the library's own (rt_comm_init_rank) for gather_frame_native: rank 0 draws the id and the
    other ranks get it through torch.distributed (whatever backend the process group runs on)."""
    ident = [lib.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ident, src=0)
    return lib.comm_init_rank(ident[0], rank, world)


def gather_frame_native(lib, comm, local, rank, world, width, height, block=ROW_BLOCK, frame=None, stream=None):
    """gather_frame through the C ABI: rt_gather_blocks sends every block straight into its place in rank 0's frame
    (grouped RCCL sends / receives issued by the library itself: no stacked tile stores, no row gather afterwards).
    Returns the [height, width, 3] frame on rank 0, None elsewhere."""
    if rank == 0 and frame is None:
        frame = torch.empty((height, width, 3), dtype=local.dtype, device=local.device)
    lib.gather_blocks(comm, 0, local.data_ptr(), frame.data_ptr() if rank == 0 else None, width, height, block, stream)
    return frame if rank == 0 else None
