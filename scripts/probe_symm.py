import os, torch, torch.distributed as dist
rank = int(os.environ.get("RANK", 0)); torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))))
try:
    import torch.distributed._symmetric_memory as symm_mem
    print("symm_mem import ok", [x for x in dir(symm_mem) if not x.startswith("_")][:40])
    t = symm_mem.empty(1024, dtype=torch.float64, device="cuda")
    hdl = symm_mem.rendezvous(t, dist.group.WORLD.group_name)
    print("rendezvous ok", type(hdl), hdl.rank, hdl.world_size, [hex(p) for p in hdl.buffer_ptrs], [hex(p) for p in hdl.signal_pad_ptrs])
    t.fill_(1.5)
    try:
        out = torch.ops.symm_mem.one_shot_all_reduce(t, "sum", dist.group.WORLD.group_name)
        print("one_shot_all_reduce f64 ok", out[:3])
    except Exception as e:
        print("one_shot_all_reduce failed:", repr(e)[:300])
except Exception as e:
    print("symm_mem failed:", repr(e)[:500])
dist.destroy_process_group()
