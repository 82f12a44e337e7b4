"""2 ranks on one GPU over gloo: step time and host profile of the DP training step with / without the
weight-gradient side stream (diagnosis of a slowdown seen in the bench self-test)."""
import cProfile, io, os, pstats, socket, sys, time
import torch

def worker(rank, world, port):
    import torch.distributed as dist
    sys.path.insert(0, "/root/repo")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from promptttspp_amd import config, functional as PF
    dev = torch.device("cuda:0"); torch.cuda.set_device(0)
    config.set_compute_dtype(torch.bfloat16)
    model = bench.build_model(dev).train()
    batches = bench.make_batches(rank, world, 6, 30000, dev)
    red, opt, sched = bench.train_setup(model, world)
    for mode in ("async", "sync"):
        PF._direct["async"] = mode == "async"
        for b in batches[:2]:
            bench.train_step(model, b, red, opt, sched)
        torch.cuda.synchronize(); dist.barrier()
        pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
        for b in batches[2:5]:
            bench.train_step(model, b, red, opt, sched)
        torch.cuda.synchronize(); pr.disable()
        dt = (time.perf_counter() - t0) / 3
        if rank == 0:
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12)
            print(f"=== {mode}: {1e3*dt:.1f} ms/step\n" + "\n".join(s.getvalue().splitlines()[6:22]), flush=True)
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = torch.multiprocessing.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, port)) for r in range(2)]
    [p.start() for p in ps]; [p.join(400) for p in ps]
