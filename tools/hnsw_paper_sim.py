"""Trace printer behind tests/golden/paper_kats.json: hnsw_dim2_m2_efc3 — a plain-Python restatement of the reference's HNSW insertion and search
(hnsw_index.go:228-288 Add, :493-552 insertNode, :565-629 searchLayer, :637-656 selectNeighbors, :667-694 pruneConnections; hnsw_index_search.go:248-354),
written from the Go source and sharing no code with oracle/comet_oracle.cpp. It only handles cases where every distance that meets another in a heap, a
sort or a greedy comparison is DISTINCT (it asserts that), so that container/heap's sift order and sort.Slice's instability cannot matter and Python's heapq
stands in for them. Squared-L2 distances of small integer points: every number in the trace can be redone on paper.

    python tools/hnsw_paper_sim.py          # prints the case's trace, graph and query results as JSON (what the fixture holds)
"""
import heapq, json, sys
def d2(a,b): return sum((x-y)**2 for x,y in zip(a,b))
class G:
    def __init__(s,M,efc,efs,log=None):
        s.M,s.efc,s.efs=M,efc,efs; s.nodes={}; s.maxLevel=-1; s.entry=0; s.log=log if log is not None else []
    def searchLayer(s,q,ep,ef,layer,tag=""):
        visited={ep}
        d=d2(q,s.nodes[ep]['v']); seen_d=[d]
        cand=[(d,ep)]; res=[(-d,ep)]
        tr=[]
        while cand:
            cd,c=heapq.heappop(cand)
            if len(res)>=ef and cd>-res[0][0]:
                tr.append(f"pop {c}(d={cd}) > worst {-res[0][0]} with {len(res)} results: stop"); break
            n=s.nodes[c]; step=[]
            if layer<len(n['e']):
                for nb in n['e'][layer]:
                    if nb in visited: step.append(f"{nb} seen"); continue
                    visited.add(nb)
                    d=d2(q,s.nodes[nb]['v']); assert d not in seen_d,('tie in search',q,nb,d); seen_d.append(d)
                    if len(res)<ef or d< -res[0][0]:
                        heapq.heappush(cand,(d,nb)); heapq.heappush(res,(-d,nb))
                        if len(res)>ef:
                            out=heapq.heappop(res); step.append(f"{nb}(d={d}) in, {out[1]}(d={-out[0]}) out")
                        else: step.append(f"{nb}(d={d}) in")
                    else: step.append(f"{nb}(d={d}) not better than worst {-res[0][0]}")
            tr.append(f"expand {c}(d={cd}): "+(", ".join(step) if step else "no edges at this layer"))
        out=sorted([(-a,b) for a,b in res])
        # distinctness check
        ds=[a for a,b in out]; assert len(set(ds))==len(ds),("tie",out)
        s.log.append({"searchLayer":tag,"layer":layer,"from":ep,"ef":ef,"steps":tr,"result":[[b,a] for a,b in out]})
        return out
    def add(s,id,v,level):
        node={'v':v,'lvl':level,'e':[[] for _ in range(level+1)]}
        if level>s.maxLevel: s.maxLevel=level
        if s.entry==0 and not s.nodes:
            s.entry=id; s.nodes[id]=node; s.log.append({"add":id,"first node: entry point":True}); return
        s.log.append({"add":id,"level":level,"maxLevel":s.maxLevel})
        curr=s.entry; cd=d2(v,s.nodes[curr]['v'])
        for lc in range(s.maxLevel,level,-1):
            ch=True
            while ch:
                ch=False; cn=s.nodes[curr]
                if lc<len(cn['e']):
                    for nb in list(cn['e'][lc]):
                        d=d2(v,s.nodes[nb]['v'])
                        assert d!=cd
                        if d<cd:
                            s.log.append({"greedy":f"layer {lc}: {curr}(d={cd}) -> {nb}(d={d})"}); cd=d; curr=nb; ch=True
        for lc in range(level,-1,-1):
            c=s.searchLayer(v,curr,s.efc,lc,f"insert {id}")
            M=s.M*(2 if lc==0 else 1)
            nbrs=[b for a,b in c][:M]
            for nb in nbrs:
                node['e'][lc].append(nb)
                n=s.nodes[nb]
                if lc<=n['lvl']:
                    n['e'][lc].append(id)
                    if len(n['e'][lc])>M:
                        old=list(n['e'][lc])
                        cl=[(d2(n['v'],s.nodes[x]['v']),x) for x in n['e'][lc] if x in s.nodes]
                        ds=[a for a,b in cl]; assert len(set(ds))==len(ds),("prune tie",nb,cl)
                        cl.sort(); n['e'][lc]=[x for _,x in cl[:M]]
                        s.log.append({"prune":nb,"layer":lc,"had":old,"distances":[[x,a] for a,x in cl],"kept":list(n['e'][lc])})
                else:
                    s.log.append({"no back link":f"{nb} has level {n['lvl']} < {lc}"})
            if c: curr=c[0][1]
        s.nodes[id]=node
    def search(s,q,k,ef):
        curr=s.entry; cd=d2(q,s.nodes[curr]['v']); path=[]
        for lc in range(s.maxLevel,0,-1):
            ch=True
            while ch:
                ch=False; n=s.nodes[curr]
                if lc<len(n['e']):
                    for nb in list(n['e'][lc]):
                        d=d2(q,s.nodes[nb]['v']); assert d!=cd
                        if d<cd: path.append(f"layer {lc}: {curr}(d={cd}) -> {nb}(d={d})"); cd=d; curr=nb; ch=True
        s.log.append({"query":q,"greedy":path,"layer-0 entry":curr})
        c=s.searchLayer(q,curr,ef if ef>0 else s.efs,0,f"query {q}")
        kk=len(c) if (k<=0 or k>len(c)) else k
        return c[:kk]


CASE = {"dim": 2, "metric": "l2_squared", "M": 2, "efConstruction": 3, "efSearch": 3,
        "insert_order": [1, 2, 3, 4, 5, 6, 7, 8, 9],
        "vectors": {"1": [5, 8], "2": [2, 6], "3": [9, 8], "4": [0, 1], "5": [3, 4], "6": [1, 1], "7": [0, 5], "8": [6, 1], "9": [6, 7]},
        "levels": {"1": 2, "2": 0, "3": 1, "4": 0, "5": 1, "6": 0, "7": 1, "8": 0, "9": 0},
        "queries": [{"q": [1, 2], "k": 3, "ef": 0}, {"q": [8, 8], "k": 2, "ef": 0}, {"q": [8, 8], "k": 0, "ef": 8}, {"q": [1, 5], "k": 4, "ef": 0},
                    {"q": [1, 2], "k": 1, "ef": 1}, {"q": [1, 5], "k": 0, "ef": 9}]}


def build_case(case=CASE):
    g = G(case["M"], case["efConstruction"], case["efSearch"])
    for i in case["insert_order"]:
        g.add(i, list(case["vectors"][str(i)]), case["levels"][str(i)])
    return g


if __name__ == "__main__":
    g = build_case()
    out = {"build_trace": g.log, "graph": {str(i): n["e"] for i, n in g.nodes.items()}, "entry": g.entry, "max_level": g.maxLevel, "queries": []}
    for q in CASE["queries"]:
        g.log = []
        r = g.search(list(q["q"]), q["k"], q["ef"])
        out["queries"].append(dict(q, ids=[b for _a, b in r], d2=[a for a, _b in r], trace=g.log))
    print(json.dumps(out, indent=1))
