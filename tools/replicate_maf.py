#!/usr/bin/env python3
"""Tiles a (synthetic) MAF along the reference chromosome: BASELINE.json configs[3] = 100 x the 1 Mb segment.

  python tools/replicate_maf.py IN.maf OUT.maf --copies 100 [--shift 1100000]

Copy k is the input with the start of every block's REFERENCE row (the first s line of the block) moved by
k * shift and the reference chromosome length raised to cover the last copy; every other byte is unchanged, so the
copies are k independent 1 Mb segments with the column content of the original (tools/gen_synthetic_maf.py makes
that: ~90 MB of text per Mb for 100 species).  Streams block by block; memory = one copy of the input.
"""
import argparse
import re


def tile(src, dst, copies, shift=1_100_000, chrom_len=None):
    """chrom_len: the reference chromosome length to write (default: twice the end of the last copy, + 200 kb); a caller that compares
    the copies with a run on one of them passes the same value to both (the frames of the - strand are counted from the chromosome's end)"""
    class A:
        pass
    args = A()
    args.src, args.dst, args.copies, args.shift, args.chrom_len = src, dst, copies, shift, chrom_len
    return _tile(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--copies", type=int, default=100)
    ap.add_argument("--shift", type=int, default=1_100_000)
    ap.add_argument("--chrom-len", type=int, default=None, help="reference chromosome length to write (default: from the last copy's end)")
    _tile(ap.parse_args())


def _tile(args):
    data = open(args.src, "rb").read()
    head_end = data.index(b"\na ") + 1 if not data.startswith(b"a ") else 0
    header, body = data[:head_end], data[head_end:]
    # per block: text up to and including "s <name> ", the start, " <size> + ", the chromosome length, the rest
    pat = re.compile(rb"(a [^\n]*\ns \S+ +)(\d+)( +\d+ +[+-] +)(\d+)( )")
    pieces, at = [], 0
    max_end = 0
    for m in pat.finditer(body):
        pieces.append((body[at:m.start()], m.group(1), int(m.group(2)), m.group(3), m.group(5)))
        size = int(m.group(3).split()[0])
        max_end = max(max_end, int(m.group(2)) + size)
        at = m.end()
    tail = body[at:]
    chrom_len = str(args.chrom_len if getattr(args, "chrom_len", None) else 2 * (max_end + (args.copies - 1) * args.shift + 100_000)).encode()
    with open(args.dst, "wb") as f:
        f.write(header)
        for k in range(args.copies):
            out = []
            for before, lead, start, mid, sp in pieces:
                out += [before, lead, str(start + k * args.shift).encode(), mid, chrom_len, sp]
            out.append(tail)
            f.write(b"".join(out))
    print("wrote %s: %d copies of %d blocks" % (args.dst, args.copies, len(pieces)))


if __name__ == "__main__":
    main()
