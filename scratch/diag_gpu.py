import sys, json, numpy as np
sys.path.insert(0,'.')
from tests.conftest import codes
import indelpost_amd as ip
from indelpost_amd.batch import JobTable
d=json.load(open('tests/golden/c_level_cases.json'))['cases']
g=ip.GpuAligner(0)
def run(cs, tag):
    ms,mm=cs[0]['match'],cs[0]['mismatch']
    g.set_scoring(ms,mm)
    jobs=JobTable.from_sequences([codes(c["read"]) for c in cs],[codes(c["ref"]) for c in cs],np.arange(len(cs),dtype=np.int32),[c["gap_open"] for c in cs],[c["gap_ext"] for c in cs],encoded=True)
    res=g.align(jobs)
    bad=[i for i,c in enumerate(cs) if res.as_dict(i)!=c['expect']]
    print(tag,"n",len(cs),"bad",bad)
    return bad,res
groups={}
for c in d: groups.setdefault((c['match'],c['mismatch']),[]).append(c)
for k,cs in groups.items():
    bad,res=run(cs,str(k))
    for i in bad[:3]:
        c=cs[i]
        print("  case",i,"len",len(c['read']),len(c['ref']),"go/ge",c['gap_open'],c['gap_ext'],"got",{k:v for k,v in res.as_dict(i).items() if k!='cigar'},"mode",int(res.records[i]['mode']),"exp",{k:v for k,v in c['expect'].items() if k!='cigar'})
        b2,_=run([c],"   single")
        b3,_=run([c]*9,"   x9")
    # repeat same batch to see determinism
    bad2,_=run(cs,str(k)+" again")
