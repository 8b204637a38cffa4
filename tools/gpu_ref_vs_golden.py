import sys,os; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from PIL import Image
from conftest import pkg, SCENES, ROOT
import torch
abi, host, device = pkg("abi"), pkg("host"), pkg("device")
sc = host.LoadedScene(SCENES["semesterbild"])
gp, _, st = device.render(sc, sc.camera, sc.settings, abi.Options.make(rng_mode=abi.RNG_REF), want_linear=False)
gold = np.array(Image.open(os.path.join(ROOT, "tests/golden/semesterbild_reference_800x600_256spp.png")).convert("RGB")).astype(np.int32)
img = np.stack([(gp >> 16) & 255, (gp >> 8) & 255, gp & 255], axis=-1).astype(np.int32)
d=np.abs(img-gold).max(-1)
print("GPU REF vs reference render: exact %.5f, <=1 %.5f, max %d, rows identical %d of 600"%((d==0).mean(), (d<=1).mean(), d.max(), (d.max(1)==0).sum()))
bad=[(y,int(np.nonzero(d[y])[0][0])) for y in range(600) if d[y].max()>0]
print("rows with a mismatch (row:first x):", bad[:40])
