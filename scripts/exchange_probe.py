import sys, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
pkg = ge.load_package()
c = pkg.Comm(pkg.comm_unique_id(), 0, 1, 0)
for nd in (18960, 99226, 589824):
    print(nd, [round(v, 1) for v in c.exchange_probe(nd, 300)])
