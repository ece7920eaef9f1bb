import os, sys, time
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import numpy as np, maf_blocks, phylocsfpp_amd as pa
G=ROOT+'/tests/golden'
m=pa.load_model(G+'/models/100vertebrates')
ctx=m.context()
for tag,f in (('small','chr22.50alignments.maf'),('medium','chr22.516alignments.maf.gz')):
    cod,off=maf_blocks.block_codons(G+'/score-msa/'+f,m.labels,G+'/models/species_aliases.tsv')
    o=np.load(G+'/oracle/%s_omega.npy'%tag); fin=np.isfinite(o)
    for host in ('0','1'):
        os.environ['PCSF_EIGEN_HOST']=host
        t=time.time(); ph,_=ctx.score_blocks(pa.OMEGA,cod,off,compute_anc=False); dt=time.time()-t
        d=np.abs(ph[fin]-o[fin]); close=np.isclose(ph[fin],o[fin],rtol=1e-6,atol=2e-6)
        print("%s host_eigen=%s: %.2f s; %d / %d rows within 1e-6; |d| median %.2e 90%% %.2e max %.3g"%(tag,host,dt,close.sum(),fin.sum(),np.median(d),np.quantile(d,0.9),d.max()),flush=True)
