"""Pr x Pc process grid and the block-cyclic index maps (ScaLAPACK conventions, row-major ranks)."""
import torch.distributed as dist


class ProcessGrid:
    def __init__(self, pr: int, pc: int, group=None):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if pr * pc != self.world:
            raise ValueError(f"process grid {pr}x{pc} does not match world size {self.world}")
        self.pr, self.pc = pr, pc
        self.r, self.c = divmod(self.rank, pc)

        # RCCL: the communicators' internal streams get HIGH priority -- they carry the panel chain (diagonal messages,
        # panels, column operands) while a machine-filling GEMM of normal priority runs on the caller's stream; a
        # collective's few workgroups should take the next free slot, not queue behind the GEMM's own waiting workgroups.
        # Whether the option exists is decided ONCE, from the signature, before any communicator is created: every rank
        # runs the same torch, so every rank makes the same number of `new_group` calls with the same arguments (a
        # try / except around the collective constructor could leave the ranks with different call counts).
        opts = None
        if dist.get_backend(group) == "nccl" and hasattr(dist, "ProcessGroupNCCL"):
            import inspect

            if "pg_options" in inspect.signature(dist.new_group).parameters and hasattr(dist.ProcessGroupNCCL, "Options"):
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
        self.high_priority_comms = opts is not None

        def new_group(ranks):
            return dist.new_group(ranks, pg_options=opts) if opts is not None else dist.new_group(ranks)

        # one communicator per process row and per process column (every rank creates all of them)
        self.row_groups = [new_group([rr * pc + cc for cc in range(pc)]) for rr in range(pr)]
        self.col_groups = [new_group([rr * pc + cc for rr in range(pr)]) for cc in range(pc)]
        # a second communicator per process column, used ONLY for the diagonal-block broadcast of the look-ahead
        # Cholesky: that broadcast is enqueued from the diagonal stream while the column exchange of the same process
        # column is enqueued from the side stream.  With one communicator both would serialise on its internal stream in
        # host issue order (correct, but the critical-path diagonal block would queue behind a bulk exchange); with two,
        # every communicator is only ever used from ONE stream and the two kinds of traffic are independent.
        self.diag_col_groups = [new_group([rr * pc + cc for rr in range(pr)]) for cc in range(pc)]
        self.world_group = group

    @staticmethod
    def default_shape(world: int):
        """2 x 4 for 8 GPUs; otherwise the most square Pr <= Pc factorisation."""
        pr = 1
        for cand in range(1, int(world ** 0.5) + 1):
            if world % cand == 0:
                pr = cand
        return pr, world // pr

    def rank_of(self, r: int, c: int) -> int:
        return r * self.pc + c

    @property
    def row_group(self):
        return self.row_groups[self.r]

    @property
    def col_group(self):
        return self.col_groups[self.c]

    @property
    def diag_col_group(self):
        return self.diag_col_groups[self.c]

    # ---- block-cyclic maps: global block index -> (owner coordinate, local block index)
    def owner_row(self, I: int) -> int:
        return I % self.pr

    def owner_col(self, J: int) -> int:
        return J % self.pc

    def local_row_blocks(self, nblocks: int, r=None):
        r = self.r if r is None else r
        return list(range(r, nblocks, self.pr))

    def local_col_blocks(self, nblocks: int, c=None):
        c = self.c if c is None else c
        return list(range(c, nblocks, self.pc))
