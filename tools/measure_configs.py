import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import indelpost_amd as ip
from indelpost_amd import synth, _lib
from indelpost_amd.batch import JobTable
from oracle import oracle as O
from oracle.oracle import cpu_batch_results, fnv1a_ops
L=_lib.lib()

def gen_windows_reads(n_windows, reads_per_window, rls, wl_lo, wl_hi, seed=synth.SEED):
    st=seed; refs=[]; reads=[]; rid=[]
    rng=np.random.default_rng(5)
    for w in range(n_windows):
        wl=int(rng.integers(wl_lo, wl_hi+1))
        ref=np.zeros(wl,np.int8); st=L.ipx_synth_window(st, ref.ctypes.data, wl); refs.append(ref)
        # split reads of this window across read lengths
        per=reads_per_window//len(rls)
        for rl in rls:
            rl=min(rl, wl)
            buf=np.zeros(per*rl,np.int8); st=L.ipx_synth_reads(st, ref.ctypes.data, wl, buf.ctypes.data, per, rl)
            reads.extend(buf.reshape(per,rl)); rid.extend([w]*per)
    return reads, refs, rid

def run(tag, jobs, scoring, check=20000):
    streams=int(os.environ.get("STREAMS","1"))
    g=(ip.MultiStreamAligner(0,*scoring,streams=streams) if streams>1 else ip.GpuAligner(0,*scoring)); g.upload(jobs)
    g.run(); g.sync()
    g.set_profiling(True)
    t0=time.perf_counter()
    for _ in range(3): g.run()
    g.sync(); dt=(time.perf_counter()-t0)/3
    kt=g.kernel_times()
    res=g.download()
    if streams==1: print('   traceback routing (bw1..7, general, wide):', g.traceback_routing())
    top=sorted(((v[0]/3,k) for k,v in kt.items()), reverse=True)
    top=[t for t in top if t[0]>=0.4]
    print("%s: n=%d  %.2f ms/step  %.2f M aln/s   top kernels: %s"%(tag, jobs.n_jobs, dt*1e3, jobs.n_jobs/dt/1e6, ", ".join("%s %.2f"%(k,t) for t,k in top)))
    print("   modes:", dict(zip(*np.unique(res.records['mode'],return_counts=True))), "flags:", dict(zip(*np.unique(res.records['flag'],return_counts=True))))
    # parity on a sample vs the CPU checker
    m=min(check, jobs.n_jobs)
    sub=jobs.shard(0,m)
    be=O.Backend("reference" if O.have_reference() else "port")
    exp=cpu_batch_results(be, sub, O.dna_matrix(*scoring), len(os.sched_getaffinity(0)))
    rec=res.records[:m]
    bad=np.zeros(m,bool)
    for f in ("score1","score2","ref_begin1","ref_end1","read_begin1","read_end1","ref_end2","cigar_len","flag"):
        bad|=rec[f]!=exp[f]
    hs=np.array([fnv1a_ops(res.cigar_ops(i)) if rec["cigar_len"][i] else 2166136261 for i in range(m)],np.uint32)
    bad|=hs!=exp["cigar_hash"]
    print("   parity vs CPU %s on first %d jobs: %d differ"%(be.kind, m, int(bad.sum())))
    g.close()

# CHUNKS=n: repeat configs 4 and 5 on n independently generated job tables (BASELINE's full sizes are 10 M and
# 9.6 M jobs: 10 and 8 chunks) and print the aggregate
chunks=int(os.environ.get("CHUNKS","1"))
which=sys.argv[1:] or ["2a","4","5"]
if "2a" in which:
    run("config2a (1,1,3,1) 1M x150 vs 300", synth.config2_jobs(1000000), (1,1))
if "4" in which:
    for ch in range(chunks):
        reads,refs,rid=gen_windows_reads(1000, 996, [75,100,125,150,200,250], 200, 600, seed=synth.SEED+7919*ch)
        jobs=JobTable.from_sequences(reads, refs, rid, 3, 1, encoded=True)
        run("config4 mixed 75-250bp, 1000 windows 200-600bp (chunk %d)"%ch, jobs, (3,2), check=20000 if chunks==1 else 5000)
if "5" in which:
    # 16 reads per locus, each with its own 300 bp window, x 6 gap settings
    nloc=12500
    for ch in range(chunks):
        reads,refs,rid=gen_windows_reads(nloc*16, 1, [150], 300, 300, seed=(synth.SEED+104729*ch))
        grid=[(3,1),(3,0),(5,1),(5,0),(4,1),(4,0)]
        R=[];I=[];GO=[];GE=[]
        for k,(r,w) in enumerate(zip(reads,rid)):
            for go,ge in grid: R.append(r); I.append(w); GO.append(go); GE.append(ge)
        jobs=JobTable.from_sequences(R, refs, I, GO, GE, encoded=True)
        run("config5 grid: %d loci x16 reads x6 penalties, per-read windows (chunk %d)"%(nloc,ch), jobs, (3,2), check=20000 if chunks==1 else 5000)
