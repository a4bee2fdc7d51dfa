"""natural-image large batch through the whole step with / without the frame balance: python tools/nat_balance_ab.py [seed_base=40000]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
W, H, NFEAT, NLINES, B, label = bench.CONFIGS[2]
for fam in ("natural", "polygons"):
    for bal in ("1", "0", "1", "0"):
        os.environ["PLF_LSD_BALANCE"] = bal
        p = bench.Pipeline(W, H, NFEAT, NLINES, B, 0, seed, family=fam)
        e, r, n = bench.timed(p, 6, 2)
        print("%-9s seed %d balance %s: %.1f fps, %.2f ms/step, region stage %.2f ms, chain %s" % (fam, seed, bal, B * 6 / e, 1e3 * e / 6, r / max(n, 1), {k: v for k, v in p.chain_stats().items() if k != "what"}), flush=True)
        p.close(); del p
