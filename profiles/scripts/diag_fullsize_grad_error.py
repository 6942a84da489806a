import importlib, sys, torch
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
import test_gpu_celeba as tce
import test_gpu_fullsize as tf
tf.setup_module(tf)
for dtype, B in (("f32", 128), ("bf16", 128), ("bf16", 8), ("f32", 8)):
    orc, G, D, tr, got, want = tce.run_steps(dtype, B, 1, seed=2, lrs=(0.0, 0.0, 0.0))
    print(dtype, B, {k: (round(got[0][k], 5), round(want[0][k], 5)) for k in got[0]}, 'G', round(tf.arena_rel_err(G, orc.G, tce.PRE_BN_BIAS), 4), 'D', round(tf.arena_rel_err(D, orc.D), 4), flush=True)
    if B == 128:
        for k, p in G.named_parameters():
            print('   ', k, round(tce.rel_err(p.grad, orc.G[k].grad), 4), float(orc.G[k].grad.norm()))
