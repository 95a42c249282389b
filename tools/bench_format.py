"""The ONE line bench.py prints on stdout, cut from the full result object.

The driver keeps a tail of stdout and parses its last line: the line has to be small (target 6 KB, tested < 12 KB) and the
last thing on stdout.  Everything bench.py measures goes to bench_full.json (and to stderr); the line carries the contract
keys, the roofline and cpu_baseline objects whole, and per measured object a few numbers: rank, seconds first / median / min,
the dominant kernel with its time and fraction of the roofline.  What the reference prints itself is what the line must
report: tools/rank.c:100-102 ("done in %.3f s rank = %d"), the rows of spasm_schur (spasm_schur.c:61-193).
"""
import json

LINE_LIMIT = 12288          # hard limit of the test; the formatter drops optional parts until the line fits
LINE_TARGET = 6144


def _r(x, digits=4):
    """numbers with a few significant digits (the full precision stays in bench_full.json)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    return x


def _short(s, n=96):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "~"


def _calls(e):
    """first / median / min of the end-to-end calls of an object"""
    if not isinstance(e, dict) or "seconds_median" not in e:
        return None
    out = {"rank": e.get("rank"), "ranks_agree": e.get("ranks_agree"), "s_first": _r(e.get("seconds_first_call")),
           "s_median": _r(e.get("seconds_median")), "s_min": _r(e.get("seconds_min")), "calls": len(e.get("seconds_all", []))}
    sp = e.get("split_of_median_call") or {}
    st = {k: _r(sp[k], 3) for k in ("pivot_search", "density_estimates", "sparse_schur", "dense_finish") if k in sp}
    if st:
        out["stages_s"] = st
    if "sparse_rounds" in sp:
        out["sparse_rounds"] = sp["sparse_rounds"]
    ev = sp.get("events") or {}
    if "factor_plans" in ev:
        out["factor_plans"] = ev["factor_plans"]
    if "cpu_rank_time" in e:
        out["cpu_rank_time"] = e["cpu_rank_time"]
    return out


def _dominant(kernels):
    """(name, ms, frac) of the slowest kernel of a {name: {ms, frac|GB_per_s}} table"""
    best = None
    for name, k in (kernels or {}).items():
        if not isinstance(k, dict) or not k.get("ms"):
            continue
        if best is None or k["ms"] > best[1]:
            frac = k.get("frac")
            if frac is None and k.get("GB_per_s"):
                frac = k["GB_per_s"] / 8000.0
            best = (name, k["ms"], frac)
    return best


def _sparse_object(o):
    """mk14.b4 / mk15.b4: the round-0 Schur complement through the default path + whole calls"""
    if not isinstance(o, dict):
        return None
    out = {"what": _short(o.get("what", ""), 64), "rows": o.get("rows"), "path": o.get("took"), "ms_per_step": _r(o.get("ms_per_step")),
           "rows_per_s": _r(o.get("rows_per_s")), "schur_nnz": o.get("schur_nnz")}
    d = (o.get("paths") or {}).get("default") or {}
    dom = _dominant(d.get("kernels"))
    if dom:
        out["dominant_kernel"] = {"name": _short(dom[0], 48), "ms": _r(dom[1]), "frac": _r(dom[2], 3)}
    if d.get("kernels"):
        out["kernels_ms"] = {_short(k, 32): _r(v.get("ms")) for k, v in d["kernels"].items() if isinstance(v, dict)}
        by = sum(v.get("algorithmic_bytes", 0) for v in d["kernels"].values() if isinstance(v, dict))
        if by and o.get("ms_per_step"):
            out["step_frac"] = _r(by / (o["ms_per_step"] * 1e-3) / 1e9 / 8000.0, 3)
    others = {k: _r(v.get("ms_per_step")) for k, v in (o.get("paths") or {}).items() if k != "default" and isinstance(v, dict) and "ms_per_step" in v}
    if others:
        out["other_paths_ms"] = others
    f = o.get("fixed_pivot_set")
    if isinstance(f, dict):
        out["fixed_pivot_set"] = {"ms_per_step": _r(f.get("ms_per_step")), "rows": f.get("rows"), "schur_nnz": f.get("schur_nnz"),
                                  "factor_image_ms": _r(f.get("factor_image_ms"))}
        if f.get("kernels_ms"):
            out["fixed_pivot_set"]["kernels_ms"] = {_short(k, 32): _r(v) for k, v in f["kernels_ms"].items()}
        if f.get("step_frac"):
            out["fixed_pivot_set"]["step_frac"] = _r(f["step_frac"], 3)
        if f.get("kernels_frac"):
            out["fixed_pivot_set"]["kernels_frac"] = {_short(k, 32): _r(v, 3) for k, v in f["kernels_frac"].items()}
    if isinstance(o.get("cpu_baseline"), dict):
        c = o["cpu_baseline"]
        out["cpu_rows_per_s"] = {"value": _r(c.get("value")), "cores": c.get("cores"), "kind": c.get("kind")}
    e = _calls(o.get("end_to_end"))
    if e:
        out["end_to_end"] = e
    return out


def compact(full):
    """the object of the stdout line (a dict) from the full result object"""
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype"):
        if k in full:
            out[k] = full[k]
    out["data"] = _short(full.get("data", ""), 96)
    cfg = dict(full.get("config") or {})
    for k in ("workload", "why_this_workload", "sharding", "note", "path"):
        if k in cfg:
            cfg[k] = _short(cfg[k], 100)
    out["config"] = cfg
    if "error" in full:
        out["error"] = _short(full["error"], 200)

    roof = full.get("roofline")
    if isinstance(roof, dict):
        r = {}
        for k, v in roof.items():
            if k in ("what_frac_is", "staged_bytes_note", "issue_bound_note"):
                continue
            if k == "kernels":
                r[k] = {_short(n, 56): {kk: _r(vv) for kk, vv in kd.items()} for n, kd in v.items()}
            elif isinstance(v, dict):
                r[k] = {kk: _r(vv) if not isinstance(vv, str) else _short(vv, 80) for kk, vv in v.items()}
            else:
                r[k] = _r(v, 6) if not isinstance(v, str) else _short(v, 64)
        out["roofline"] = r
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        c = {k: (_r(v) if not isinstance(v, str) else _short(v, 100)) for k, v in cb.items() if not isinstance(v, dict)}
        if isinstance(cb.get("rank_time"), dict):
            c["rank_time"] = {k: (_r(v) if not isinstance(v, str) else _short(v, 100)) for k, v in cb["rank_time"].items()}
        out["cpu_baseline"] = c
    for k in ("factor_image_ms", "rows_per_s_cold"):
        if k in full:
            out[k] = _r(full[k])

    summary = {}
    e = _calls(full.get("end_to_end"))
    if e:
        h = (full.get("end_to_end") or {}).get("with_the_pivot_search_on_the_host")
        if isinstance(h, dict):
            e["s_with_host_pivot_search"] = _r(h.get("seconds"))
        summary["end_to_end"] = e
    rb = full.get("row_by_row_path")
    if isinstance(rb, dict):
        summary["row_by_row_path"] = {"ms_per_step": _r(rb.get("ms_per_step")), "kernel": _short(rb.get("kernel", ""), 40), "kernel_ms": _r(rb.get("kernel_ms")),
                                      "effective_frac": _r(rb.get("frac"), 3), "hbm_frac": _r(rb.get("hbm_frac"), 3), "same_nnz": rb.get("same_nnz")}
    dt = full.get("dense_tail")
    if isinstance(dt, dict):
        summary["dense_tail"] = {"shape": dt.get("shape"), "rank": dt.get("rank"), "ms": _r(dt.get("ms")), "update_kernels_ms": _r(dt.get("update_kernels_ms_serialised")),
                                 "mfma_busy_pct": _r(dt.get("mfma_busy_pct"), 3), "mfma_busy_source": dt.get("mfma_busy_source"),
                                 "frac_of_i8_peak": _r(dt.get("mfma_i8_frac_of_peak"), 3)}
    dr = full.get("dense_tail_real")
    if isinstance(dr, dict) and "shape" in dr:
        summary["dense_tail_real"] = {"shape": dr.get("shape"), "rank": dr.get("rank"), "ms_first": _r(dr.get("ms_first")), "ms": _r(dr.get("ms_median_of_the_rest"))}
    for key in ("at_scale", "sparse_path"):
        s = _sparse_object(full.get(key))
        if s:
            summary[key] = s
    si = full.get("stand_ins")
    if isinstance(si, list):
        rows = []
        for s in si:
            c = _calls(s) or {}
            c.pop("ranks_agree", None) if c.get("ranks_agree") else None
            rows.append(dict({"name": s.get("name"), "for": _short(s.get("stand_in_for", ""), 28), "options": s.get("options")}, **c))
        summary["stand_ins"] = rows
    b = _calls(full.get("beyond_scale"))
    if b:
        summary["beyond_scale"] = b
    dp = full.get("dist_product_path")
    if isinstance(dp, dict):
        summary["dist_product_path"] = {k: (_r(v) if not isinstance(v, str) else _short(v, 100)) for k, v in dp.items() if not isinstance(v, (dict, list))}
    cf = full.get("configs")
    if isinstance(cf, list):
        summary["configs"] = {c.get("name"): c.get("status") for c in cf}
    out["summary"] = summary
    out["full"] = full.get("full_path", "bench_full.json")
    return out


def line(full):
    """the stdout line: compact(full) as JSON; optional parts are dropped, the least important first, while it exceeds the target"""
    obj = compact(full)
    text = json.dumps(obj, separators=(",", ":"))
    drops = [("summary", "stand_ins", "stages_s"), ("summary", "sparse_path", "kernels_ms"), ("summary", "at_scale", "kernels_ms"),
             ("summary", "configs"), ("summary", "row_by_row_path"), ("summary", "stand_ins"), ("summary", "sparse_path"),
             ("summary", "beyond_scale"), ("summary", "at_scale"), ("summary",)]
    for path in drops:
        if len(text) <= LINE_TARGET:
            break
        node = obj
        ok = True
        for k in path[:-1]:
            node = node.get(k) if isinstance(node, dict) else None
            if node is None:
                ok = False
                break
        if not ok:
            continue
        if isinstance(node, list):
            for item in node:
                if isinstance(item, dict):
                    item.pop(path[-1], None)
        elif isinstance(node, dict):
            node.pop(path[-1], None)
        text = json.dumps(obj, separators=(",", ":"))
    return text
