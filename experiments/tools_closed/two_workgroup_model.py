# Discrete-event model of two co-resident kz_board_conv_f16 workgroups sharing one CU's MFMA pipe (phase lengths from the
# s_memtime stamps): reproduces the measured 80k-cycle lifetime and shows what priority / stagger / faster phases buy.
import random
# phases: (kind, amount); kind 'n' = non-MFMA time (cycles), 'k' = MFMA work (pipe-cycles)
def wg_phases(jit=0.0):
    j=lambda x: x*(1+random.uniform(-jit,jit))
    ph=[('n',j(9500))]
    for c in range(4):
        ph.append(('n',j(3000))); ph.append(('k',6900))
    ph.append(('n',j(8000)))
    return ph
def sim(nwg=400, alone_rate=0.78, prio=False, lag0=0, jit=0.05, gap=1200):
    t=0.0; slots=[None,None]; done=0; started=0
    # slot state: [phases, idx, remaining]
    def new():
        nonlocal started
        started+=1
        ph=wg_phases(jit); return [ph,0,ph[0][1]]
    slots[0]=new(); slots[1]=new(); slots[1][2]+=lag0
    wait=[0,0]
    while done<nwg:
        # rates
        ink=[wait[i]<=0 and s[0][s[1]][0]=='k' for i,s in enumerate(slots)]
        rate=[0,0]
        for i,s in enumerate(slots):
            if wait[i]>0: rate[i]=None; continue
            if s[0][s[1]][0]=='n': rate[i]=1.0
            else:
                if ink[0] and ink[1]:
                    if prio: rate[i]= alone_rate if i==0 else (1-alone_rate)
                    else: rate[i]=0.5
                else: rate[i]=alone_rate
        # time to next event
        dts=[]
        for i,s in enumerate(slots):
            if wait[i]>0: dts.append(wait[i])
            else: dts.append(s[2]/rate[i])
        dt=min(dts); t+=dt
        for i,s in enumerate(slots):
            if wait[i]>0:
                wait[i]-=dt
                if wait[i]<=1e-9: wait[i]=0; slots[i]=new()
            else:
                s[2]-=dt*rate[i]
                if s[2]<=1e-9:
                    s[1]+=1
                    if s[1]>=len(s[0]): done+=1; wait[i]=gap
                    else: s[2]=s[0][s[1]][1]
    return t/done*2  # avg lifetime-equivalent per WG per slot
for args in [dict(),dict(lag0=8000),dict(prio=True),dict(jit=0.3),dict(alone_rate=0.95),dict(alone_rate=0.95,lag0=8000)]:
    random.seed(1); print(args, round(sim(**args)))
