"""The N>1 form of the training step on ONE GPU: a 1-rank RCCL group with glass_amd.dist told to treat it as
distributed, so TrainStep takes its split form (captured forward/backward, eager RCCL exchange of the gradient arena
with ReduceOp.AVG, eager fused Adam).  Two scenarios: use_deg-style features (one small bucket: all-reduce) and
use_nodeid-style features with the embedding table forced into the big bucket (two graphs cut at the tail hook, small
all-reduce on the communication stream beside the tail, reduce-scatter + sharded Adam + all-gather for the table).
Prints the parameter hash after 20 steps next to the single-process form's: with one rank the average is the identity
and the shard is the whole bucket, so they must be equal.  Used by tests/test_gpu_model.py."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as td


def run(dist_mode, nodeid=False, capture_collective=True, force_mismatch=False, force_capture_failure=False, head=False):
    from glass_amd import synth, losses, ops, dist as gdist, step as step_mod
    step_mod.CAPTURE_COLLECTIVE = capture_collective
    from glass_amd.arena import ParamArena
    from glass_amd.optim import FlatAdam
    from glass_amd.step import TrainStep
    from glass_amd.factory import build_glass
    dev = torch.device("cuda", 0)
    if dist_mode:
        gdist.is_distributed = lambda: True  # a 1-rank group: world_size() == 1, rank() == 0
    w, ei, ew, x, pos, y = synth.make_workload("tiny", seed=0, n_batches=4)
    ei, ew, x, pos, y = (torch.from_numpy(a).to(dev) for a in (ei, ew, x, pos, y))
    if nodeid:
        x = torch.arange(x.shape[0], device=dev).reshape(-1, 1, 1)  # use_nodeid: V = N, identity gather
    torch.manual_seed(0)
    ops.rng_seed(7, dev)
    model = build_glass(64, w.layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=0.5).to(dev).train()
    arena = ParamArena(model, big_elems=4096 if nodeid else 1 << 18)  # 300 x 64 table -> the big bucket
    assert (arena.big_start < arena.flat.numel()) == nodeid
    opt = FlatAdam(arena, lr=1e-2)
    step = TrainStep(model, opt, losses.CrossEntropy(), x, ei, ew, arena, use_graph=True, warmup_iters=2, preserve_state=True)
    step._force_verify_mismatch = force_mismatch  # the replay-vs-eager check of a captured collective reports a mismatch
    step._force_capture_failure = force_capture_failure  # the capture attempt itself fails: agreed on by all ranks, split form
    B = w.batch
    if head:  # round 6: the labels inside the step's head launch, the batch named by the device cursor (TrainStep.begin_epoch)
        assert step.begin_epoch(pos, y, torch.arange(4 * B, device=dev).reshape(4, B), wrap=True)
        for k in range(20):
            step.next_step()
        assert int(step._labels.cursor[5]) == 20, int(step._labels.cursor[5])
    else:
        for k in range(20):
            b = k % 4
            step(pos[b * B:(b + 1) * B], y[b * B:(b + 1) * B])
    torch.cuda.synchronize()
    # one small bucket: the exchange + Adam are captured with the step when RCCL allows it (else the split form);
    # with an embedding-sized bucket: two graphs + the small all-reduce beside the backward tail, collectives eager
    assert step.graphed and (step._split or step.collective_in_graph) == bool(dist_mode)
    assert not (step.collective_in_graph and not capture_collective)
    assert (step._g_tail is not None) == bool(dist_mode and nodeid)
    form = "one-graph" if step.collective_in_graph else ("split" if step._split else "single")
    if step.collective_in_graph:  # a captured collective is only kept after one replay reproduced an eager step
        assert step.capture_verified and step.capture_verified["ok"], step.capture_verified
    share = step.collective_share()
    return (hashlib.md5(arena.flat_param.cpu().numpy().tobytes()).hexdigest(), form, step.capture_error, step.capture_verified,
            share["payload_bytes"] if share else None)


if __name__ == "__main__":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    single = {nid: run(False, nid) for nid in (False, True)}
    head_single = run(False, False, head=True)
    td.init_process_group("nccl", device_id=torch.device("cuda", 0))
    split = {nid: run(True, nid) for nid in (False, True)}
    head_split = {nid: run(True, nid, head=True) for nid in (False, True)}
    eager_coll = run(True, False, capture_collective=False)  # the split form, whatever the capture attempt above did
    opted_out = run(True, False, force_mismatch=True)         # capture succeeds, the check "fails": automatic opt-out
    refused = run(True, False, force_capture_failure=True)    # the capture attempt fails: outcome agreed (all-reduce MIN), split form
    td.barrier(device_ids=[0])
    td.destroy_process_group()
    ok = True
    for nid in (False, True):
        same = single[nid][0] == split[nid][0]
        ok = ok and same
        print("nodeid" if nid else "deg", "single", single[nid][0], split[nid][1], split[nid][0],
              "same" if same else "DIFFERENT", "capture_error:", split[nid][2])
    same = single[False][0] == head_single[0]
    ok = ok and same
    print("deg single", single[False][0], "labels in the head launch:", head_single[0], "same" if same else "DIFFERENT")
    for nid in (False, True):
        same = single[nid][0] == head_split[nid][0] and head_split[nid][1] == split[nid][1]
        ok = ok and same
        print("nodeid" if nid else "deg", "single", single[nid][0], "head launch +", head_split[nid][1], head_split[nid][0],
              "same" if same else "DIFFERENT", "verified:", head_split[nid][3], "capture_error:", head_split[nid][2])
    same = single[False][0] == eager_coll[0]
    ok = ok and same and eager_coll[1] == "split"
    print("deg single", single[False][0], eager_coll[1], eager_coll[0], "same" if same else "DIFFERENT")
    same = single[False][0] == opted_out[0]
    # (if the runtime refused the capture in the first place there was nothing to verify: the split form is taken anyway)
    ok = ok and same and opted_out[1] == "split" and (opted_out[3] is None or "replay-vs-eager" in str(opted_out[2]))
    print("deg single", single[False][0], "opt-out after a forced mismatch:", opted_out[1], opted_out[0],
          "same" if same else "DIFFERENT", "verified:", opted_out[3], "| trusted capture:", split[False][3])
    same = single[False][0] == refused[0]
    ok = ok and same and refused[1] == "split" and "forced capture failure" in str(refused[2])
    print("deg single", single[False][0], "capture refused on this rank:", refused[1], refused[0], "same" if same else "DIFFERENT",
          "capture_error:", refused[2])
    # what the exchange of this model should cost at N = 2, 4, 8 (glass_amd.dist.predict_collective_us: a stated model with
    # assumed constants) — the figure a SCALE run's collective.exposed_us is held against
    from glass_amd import dist as gdist
    for nid in (False, True):
        pay = split[nid][4]
        print("predicted exchange", "nodeid" if nid else "deg", pay,
              {n: round(gdist.predict_collective_us(pay, n)["total_us"], 1) for n in (2, 4, 8)}, "us at N = 2 / 4 / 8")
    print("ALL EQUAL" if ok else "MISMATCH")
