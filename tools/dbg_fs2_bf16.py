import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
import test_gpu_fs2 as t
dev=torch.device("cuda:0")
ref, model = t._models(t.FastSpeech2ConfigRef(), dev, seed=3)
ids, lens, g = t._batch(80, 2, 40, seed=11, lens=[40, 23])
durs = torch.randint(2, 9, (2, 40), generator=g)
want = ref(ids, lens, durations=durs)
for prec in ("f32","bf16"):
    model.precision = prec
    got = model(ids, lens, durations=durs)
    for i,name in ((3,"pitch"),(4,"energy"),(0,"mel"),(1,"postnet")):
        e=(got[i].cpu()-want[i])
        print(prec, name, "relL2 %.3e median %.3e max %.3e scale %.3e" % (float(e.norm()/want[i].norm()), float(e.abs().median()), float(e.abs().max()), float(want[i].abs().max())))
