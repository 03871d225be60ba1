import numpy as np
G=np.load("tests/golden/train_step_aux.npz")
for ph in ("warmup","full"):
    for net in ("irradiance_mlp","roughness_mlp","albedo_mlp","normal_mlp"):
        row=[]
        for l in range(8):
            b=G["%s__grad_%s__positions_linears.%d.bias"%(ph,net,l)]
            row.append("%.1e"%np.abs(b).max())
        w={l:"%.1e"%np.abs(G["%s__grad_%s__positions_linears.%d.weight"%(ph,net,l)]).max() for l in (0,5,7)}
        print(ph,net,"bias max per layer",row,"weights",w)
