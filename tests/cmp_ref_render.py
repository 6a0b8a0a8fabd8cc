import sys, json, numpy as np, os
ROOT='/root/repo' if os.path.exists('/root/repo/tests') else os.getcwd()
sys.path.insert(0, ROOT)
import gdpt_amd as G
d=json.load(open(os.path.join(ROOT,'tests/golden/ref_images.json')))['reference_renders']
sc=G.Scene(G.parse_scene(os.path.join(ROOT,'scenes/cbox/cbox_gdpt.xml')))
for spp in (1,4,16,64):
    out=sc.gradient_path_render(spp, G.RNG_SAMPLE, alpha=0.04)
    h,w,_=out.shape; bs=32
    th=out.reshape(h//bs,bs,w//bs,bs,3).mean(axis=(1,3))
    print("spp",spp,"mean",out.mean(axis=(0,1)),"neg",float((out<0).any(axis=2).mean()), "pct", np.percentile(out.mean(axis=2),[1,50,99]))
    for k,v in d.items():
        ref=np.array(v['block_mean_32'])
        print("   vs",k,"relL2 thumb", np.linalg.norm(th-ref)/np.linalg.norm(ref), "mean ratio", out.mean(axis=(0,1))/np.array(v['mean']))
