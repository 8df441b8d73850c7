import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import *  # noqa
from test_gpu_parity import golden_inputs
from diffusionremotesensing_amd import synthetic
from diffusionremotesensing_amd.UNet_model_superres import Residual_Attention_UNet_superres
dev = torch.device("cuda:0")
m = Residual_Attention_UNet_superres(3, 3, dev)
m.load_state_dict(synthetic.seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, t, lr = golden_inputs("cfg2", 16, 16, 3, 256, 2, 1500)
with torch.no_grad():
    got = m(x.to(dev), t.to(dev), lr.to(dev), 2).cpu()
    for i in (0, 5, 15):
        single = m(x[i:i+1].to(dev), t[i:i+1].to(dev), lr[i:i+1].to(dev), 2).cpu()
        dlt = (got[i:i+1] - single).abs()[0].amax(0)
        ys, xs = torch.nonzero(dlt > 1e-7, as_tuple=True)
        print(i, "max", float(dlt.max()), "count", len(ys), "rows", ys.min().item() if len(ys) else None, ys.max().item() if len(ys) else None,
              "cols", xs.min().item() if len(xs) else None, xs.max().item() if len(xs) else None)
        if len(ys):
            print("  row mod 4 hist", torch.bincount(ys % 4, minlength=4).tolist(), "col mod 16 hist", torch.bincount(xs % 16, minlength=16).tolist())
    print("checksums: batch image 5 %.9f  single image 5 %.9f" % (float(got[5].double().sum()),
          float(m(x[5:6].to(dev), t[5:6].to(dev), lr[5:6].to(dev), 2).double().sum())))
    g2 = m(x.to(dev), t.to(dev), lr.to(dev), 2).cpu()
    print("batch run-to-run max diff", float((g2 - got).abs().max()))
    s1 = m(x[5:6].to(dev), t[5:6].to(dev), lr[5:6].to(dev), 2).cpu()
    s2 = m(x[5:6].to(dev), t[5:6].to(dev), lr[5:6].to(dev), 2).cpu()
    print("single run-to-run max diff", float((s2 - s1).abs().max()))
    for nb in (2, 4, 8):
        gb = m(x[:nb].to(dev), t[:nb].to(dev), lr[:nb].to(dev), 2).cpu()
        print("batch", nb, "image 0 vs batch-16 image 0:", float((gb[0] - got[0]).abs().max()))
