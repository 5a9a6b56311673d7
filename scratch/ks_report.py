import sys
sys.path.insert(0,'scratch')
from dbtrace import rows
def rep(path):
    r=[x for x in rows(path) if 'igemm_kernel' in x[0]]
    i=0; out={}
    for flush in (1,0):
        for m,n in [(4608,640),(18432,320),(1152,1280)]:
            for k in (64,320,640,1280,2560):
                for res in (0,1):
                    seg=r[i:i+3]; i+=3
                    out[(flush,m,n,k,res)]=min(s[4] for s in seg)
    return out
a=rep(sys.argv[1]); b=rep(sys.argv[2])
for key in a:
    flush,m,n,k,res=key
    print("flush=%d M=%5d N=%4d K=%4d res=%d   %6.1f -> %6.1f us   TF %4.0f -> %4.0f"%(flush,m,n,k,res,a[key],b[key],2*m*n*k/a[key]/1e6,2*m*n*k/b[key]/1e6))
