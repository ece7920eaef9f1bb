"""TEST INFRASTRUCTURE: plain-Python restatement of the reference's smoothing pass, used only to check the C++
host implementation (phylocsfpp_amd/host/hmm.hpp).  Written from the reference's behaviour:
  get_coding_hmm                      src/create_tracks.hpp:163-200
  state_posterior_probabilities       src/create_tracks.hpp:90-159
  get_best_path_by_viterbi / indexmax src/create_tracks.hpp:29-88
  compute_log_odds, process_scores    src/create_tracks.hpp:226-314
  estimate_hmm_params_for_genome      src/estimate_hmm_parameter.hpp:38-340
  wig_reader::get_next_scores         src/wig_file_reader.hpp:91-140
Parity with the reference's smoothed golden files is pinned only to a tolerance (tests/golden/fit_hmm_params.py)."""
import math


def make_hmm(p):
    """p = [coding_prior, coding_length, w0, w1, w2, L0, L1, L2] -> (init[4], trans[4][4])"""
    prior, lc, w, ln = p[0], p[1], p[2:5], p[5:8]
    un = [w[i] * ln[i] for i in range(3)]
    c2n = [w[i] / lc for i in range(3)]
    n2c = [1.0 / ln[i] for i in range(3)]
    init = [prior] + [(1 - prior) * un[i] / (un[0] + un[1] + un[2]) for i in range(3)]
    trans = [[0.0] * 4 for _ in range(4)]
    trans[0][0] = 1.0 - (c2n[0] + c2n[1] + c2n[2])
    for j in range(3):
        trans[0][j + 1] = c2n[j]
    for i in range(1, 4):
        trans[i][0] = n2c[i - 1]
        trans[i][i] = 1.0 - n2c[i - 1]
    return init, trans


def emit(state, score):
    return 10.0 ** (score / 10.0) if state == 0 else 1.0


def coding_posterior(init, trans, obs):
    n = len(obs)
    fwd = [[0.0] * 4 for _ in range(n)]
    bwd = [[1.0] * 4 for _ in range(n)]
    for s in range(4):
        fwd[0][s] = init[s] * emit(s, obs[0])
    for pos in range(1, n):
        for s in range(4):
            tot = 0.0
            for q in range(4):
                tot += fwd[pos - 1][q] * trans[q][s]
            fwd[pos][s] = tot * emit(s, obs[pos])
        m = max(fwd[pos])
        fwd[pos] = [v / m for v in fwd[pos]]
    for pos in range(n - 2, -1, -1):
        for s in range(4):
            tot = 0.0
            for q in range(4):
                tot += trans[s][q] * emit(q, obs[pos + 1]) * bwd[pos + 1][q]
            bwd[pos][s] = tot
        m = max(bwd[pos])
        bwd[pos] = [v / m for v in bwd[pos]]
    post = []
    for pos in range(n):
        total = 0.0
        for s in range(4):
            total += fwd[pos][s] * bwd[pos][s]
        post.append(fwd[pos][0] * bwd[pos][0] / total)
    return post


def log_odds(prob):
    if prob < 10.0 ** -15:
        return -15.0
    if prob > 1 - 10.0 ** -15:
        return 15.0
    return math.log10(prob / (1 - prob))


def viterbi(init, trans, obs):
    n = len(obs)
    prev = [init[s] * emit(s, obs[0]) for s in range(4)]
    back = []
    for pos in range(1, n):
        cur, arg = [0.0] * 4, [0] * 4
        for s in range(4):
            best, a = 0.0, 0
            for q in range(4):
                v = prev[q] * trans[q][s]
                if v > best:
                    best, a = v, q
            cur[s] = best * emit(s, obs[pos])
            arg[s] = a
        m = max(cur)
        prev = [v / m for v in cur]
        back.append(arg)
    best = 0
    for i in range(1, 4):
        if prev[i] > prev[best]:
            best = i
    path = [best]
    for arg in reversed(back):
        path.append(arg[path[-1]])
    path.reverse()
    return path


def regions(path, post, block_start):
    """list of (start, end, prob) with the reference's boundary rules"""
    out = []
    if len(path) < 2:
        return out
    start_pos = start_cnt = 0
    last = len(path) - 2
    for i in range(len(path) - 1):
        if i == 0 and path[i] == 0:
            start_pos, start_cnt = block_start - 1, 0
            if path[i + 1] != 0:
                out.append((start_pos, start_pos + 3, post[0]))
        elif path[i + 1] == 0 and path[i] != 0:
            if i != last:
                start_pos, start_cnt = block_start + 3 * i + 2, i + 1
            else:
                end = block_start + 3 * i + 5
                out.append((end - 3, end, post[i + 1]))
        elif path[i + 1] != 0 and path[i] == 0:
            out.append((start_pos, block_start + 3 * i + 2, max([0.0] + post[start_cnt:i + 1])))
        elif i == last and path[i + 1] == 0 and path[i] == 0:
            out.append((start_pos, block_start + 3 * i + 5, max([0.0] + post[start_cnt:i + 2])))
    return out


def wig_runs(path):
    """[(chrom, start, [values])], contiguous fixedStep sections merged"""
    runs = []
    for line in open(path):
        if line.startswith("f"):
            chrom = line.split("chrom=")[1].split()[0]
            start = int(line.split("start=")[1].split()[0])
            if runs and runs[-1][2] and runs[-1][0] == chrom and runs[-1][1] + 3 * len(runs[-1][2]) == start:
                continue
            runs.append((chrom, start, []))
        elif line.strip():
            runs[-1][2].append(float(line))
    return [r for r in runs if r[2]]


# ---- parameter estimation ----------------------------------------------------------------------------------
def _nll(points, weight, x):
    f = 0.0
    tau = 10.0 ** x
    for p, w in zip(points, weight):
        f -= w * (-float(p) / tau - math.log(tau))
    return f


def _minimise_1d(points, weight, guess, scale, rel_tol):
    sx = sorted([(guess, _nll(points, weight, guess)), (guess + scale, _nll(points, weight, guess + scale))], key=lambda v: v[1])
    xtol = rel_tol * scale
    prev_er = True
    for _ in range(31):
        if not prev_er:
            xs = [v[0] for v in sx]
            if max(max(xs), 2.2250738585072014e-308) - min(xs) < xtol:
                return sx[0][0]
        centroid = sx[0][0]
        refl = centroid + (centroid - sx[1][0])
        f_ref = _nll(points, weight, refl)
        er = False
        if sx[0][1] <= f_ref < sx[0][1]:
            sx[1] = (refl, f_ref)
        elif f_ref < sx[0][1]:
            exp_ = centroid + 2 * (centroid - sx[1][0])
            f_exp = _nll(points, weight, exp_)
            if f_exp < f_ref:
                sx[1] = (exp_, f_exp)
                er = True
            else:
                sx[1] = (refl, f_ref)
        else:
            con = centroid - .5 * (centroid - sx[1][0])
            f_con = _nll(points, weight, con)
            if f_con < sx[1][1]:
                sx[1] = (con, f_con)
            else:
                er = True
                x = sx[0][0] + 0.5 * (sx[1][0] - sx[0][0])
                sx[1] = (x, _nll(points, weight, x))
        sx.sort(key=lambda v: v[1])
        prev_er = er
    raise RuntimeError("Nelder-Mead did not converge")


def estimate_params(exon_lines, genome_length):
    """exon_lines: iterable of 'chrom strand frame start end'; valid for <= 20000 gaps (no subsampling)"""
    by_key = {}
    for line in exon_lines:
        t = line.split()
        if len(t) < 5:
            continue
        by_key.setdefault(":".join(t[:3]), []).append((int(t[3]), int(t[4])))
    num_exons = coding_nt = 0
    gaps = []
    for key in sorted(by_key):
        lst = sorted(by_key[key])
        i = tail = 0
        head = 1
        while i < len(lst) - 1:
            if lst[head][0] <= lst[tail][1]:
                if lst[tail][1] - lst[tail][0] >= lst[head][1] - lst[head][0]:
                    del lst[head]
                else:
                    del lst[tail]
                    head = tail + 1
            else:
                head += 1
                tail += 1
                i += 1
        k = 0
        while k < len(lst):  # steps two exons at a time
            end1 = lst[k][1]
            k += 1
            if k == len(lst):
                break
            if lst[k][0] > end1 + 1:
                gaps.append(lst[k][0] - end1 - 1)
            k += 1
        num_exons += len(lst)
        coding_nt += sum(e - s + 1 for s, e in lst)
    assert len(gaps) <= 20000
    x = [math.log10(3000.0), math.log10(80000.0), math.log10(100.0)]
    prior = [30.0 / 41.0, 10.0 / 41.0, 1.0 / 41.0]
    n = len(gaps)
    for _ in range(20):
        resp = [[0.0] * n for _ in range(3)]
        for i, g in enumerate(gaps):
            lik = []
            for j in range(3):
                tau = 10.0 ** x[j]
                lik.append(prior[j] * math.exp(-float(g) / tau - math.log(tau)))
            total = lik[0] + lik[1] + lik[2]
            for j in range(3):
                resp[j][i] = lik[j] / total if total != 0.0 else 1.0 / 3
        for j in range(3):
            s = 0.0
            for v in resp[j]:
                s += v
            prior[j] = s / float(n)
        for j in range(3):
            if x[j] == 0:
                continue
            x[j] = _minimise_1d(gaps, resp[j], x[j], 0.1, 0.001)
    return [coding_nt / float(genome_length) / 6.0, coding_nt / float(num_exons) / 3.0] + prior + [10.0 ** v / 3 for v in x]
