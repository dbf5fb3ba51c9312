"""Child process of tests/test_ddp_gpu.py::test_folded_layerscale_under_accumulation_and_sharded_reducer: ONE process, a gloo
group of world size 1 with the collectives FORCED (the sharded reducer's reduce-scatter path really runs), two accumulated
micro-batches; saves the flat gradient.  The parent runs it with VLM_FOLD_LAYERSCALE=1 and =0 (the switch is read at import).

usage: fold_accumulate.py OUT.npz      env: MASTER_ADDR, MASTER_PORT
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ddp_one_device as H  # noqa: E402


def main():
    out = sys.argv[1]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    H.ge.import_package()
    ddp = importlib.import_module("vl_merging_amd.ddp")
    model, vm = H.build_model()
    (opt,), (sch,) = vm.vilt_utils.set_schedule(model, max_steps=100)
    red = ddp.FlatGradReducer(model, sharded=True, force_collectives=True).attach(opt, defer_tail=False)
    nb = H.fixed_mask_batch(4)
    for it in range(2):  # the first step teaches the reducer its use counts; the second is the one that is compared
        opt.zero_grad()
        for micro in range(2):  # run.py's accumulation protocol: only the last micro-batch's backward sends gradients
            red.accumulate = micro == 0
            red.begin_step()
            loss = model.training_step({"vl": H.gpu_rows(nb, 2 * micro, 2 * micro + 2)}, it)
            loss.backward()
            red.finish_backward()
    torch.cuda.synchronize()
    f = model._flat
    assert not f.ls_pending if hasattr(f, "ls_pending") else True
    np.savez(out, grad=f.flat_g[:f.numel].cpu().numpy(), names=np.array(f.names), offsets=np.array([f.offsets[n][0] for n in f.names]))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
