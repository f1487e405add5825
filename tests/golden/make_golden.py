#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs only in the dev container, where /root/reference is mounted.  It copies the
reference checkout to a scratch directory (the mount is read-only), compiles the
reference's single C file there exactly as its setup.py would (one shared object named
kmers_c<EXT_SUFFIX> beside kmers.py), imports the reference's own Python modules from
that scratch copy and records inputs + outputs as JSON.  Nothing from the scratch copy
is written into this repository: the fixtures are data (inputs and expected outputs).

    python tests/golden/make_golden.py

Fixtures written (see SURVEY.md §8c, F1–F6):
    kat.json            reference unit KATs re-evaluated through the real reference
    toy_cli.json        CLI runs on the reference's toy inputs: TSV, counts, bins
    diff_vectors.json   seeded differential vectors, k in {5,13,21,27,31,32}
    edge_vectors.json   priority / short-read / duplicate / no-trailing-newline cases
    readfq_vectors.json parser quirks of readfq + Read.print formats
    readfq_fuzz.json    400 random byte strings over the characters that steer the parser
    cli_misc.json       output-extension rule and float formatting
"""
import gzip
import hashlib
import json
import os
import random
import shutil
import subprocess
import sys
import sysconfig
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
DATA_OUT = os.path.join(os.path.dirname(HERE), "data")


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def setup_reference(tmp):
    ref = os.path.join(tmp, "ref")
    shutil.copytree(REF, ref)
    so = os.path.join(ref, "src", "trio_binning", "kmers_c" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-w", "-o", so, os.path.join(ref, "c", "kmers.c")], check=True)
    return ref


def run_cli(ref, cwd, argv):
    """Run the reference CLI in a child interpreter so its never-closed output files are
    finalised by interpreter shutdown (classify_by_kmers.py:80-117 has no close)."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.argv = %r;"
        "from trio_binning.classify_by_kmers import main; main()"
        % (os.path.join(ref, "src"), ["classify-by-kmers"] + argv)
    )
    p = subprocess.run([sys.executable, "-c", code], cwd=cwd, capture_output=True, check=True)
    return p.stdout.decode()


COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def canonical(s, kmer_to_int):
    r = rc(s)
    return s if kmer_to_int(s) <= kmer_to_int(r) else r


def rand_seq(rng, n):
    return "".join(rng.choice("ACGT") for _ in range(n))


def main():
    if not os.path.isdir(REF):
        sys.exit("reference checkout not mounted; fixtures can only be regenerated in the dev container")
    tmp = tempfile.mkdtemp(prefix="tbk_golden_")
    ref = setup_reference(tmp)
    sys.path.insert(0, os.path.join(ref, "src"))
    from trio_binning import kmers as rk  # the reference's module
    from trio_binning import seq as rs
    import io

    rdata = os.path.join(ref, "tests", "data")

    # Toy inputs are test DATA held by the reference's own tests (MIT-licensed); copy
    # them as fixtures so the GPU box has them.
    os.makedirs(DATA_OUT, exist_ok=True)
    for name in ("hapA.txt", "hapB.txt", "test.fa", "test.fastq", "test.ccs.fastq.gz", "hapA.fastq"):
        shutil.copyfile(os.path.join(rdata, name), os.path.join(DATA_OUT, name))

    # ---- F1 / F3: unit KATs ----------------------------------------------------------
    kat_kmers = [
        "ATGCTAGCTAGAGAGAGAGGA", "T" * 28, "TTTTTTTTTTTTTTTAGGCCCACTTTTT", "A" * 26,
        "GGGAGGGAGGGAGGGAGGGAGGGAGGG", "C", "AC", "CA", "ACGT", "N", "acgt", "T" * 32, "A" * 32,
        "ACGTACGTACGTACGTACGTACGTACGTACGT", "GATTACA",
    ]
    kat = {
        "kmer_to_int": [[s, rk.kmer_to_int(s)] for s in kat_kmers],
        "reverse_complement": [[s, rk.reverse_complement(s)] for s in kat_kmers if set(s) <= set("ACGT")],
        "hash_function_note": "hash values are not observable; none recorded",
    }
    hapA = rk.create_kmer_hash_set(os.path.join(rdata, "hapA.txt"))
    hapB = rk.create_kmer_hash_set(os.path.join(rdata, "hapB.txt"))
    read72 = "CTTATCATGTCTTTGTTTTCAAAGCTTCTTAGAGGTTTTTTTTTTTGGTGTTAATTGGCATAAATTATGGCT"
    kat["tables"] = {
        "hapA.txt": {"k": hapA.contents.k, "num_kmers": rk.get_number_kmers_in_set(hapA)},
        "hapB.txt": {"k": hapB.contents.k, "num_kmers": rk.get_number_kmers_in_set(hapB)},
    }
    kat["count_kmers_in_read"] = [
        {"read": read72, "counts": list(rk.count_kmers_in_read(read72, hapA, hapB))},
        {"read": "GAGGAGATTTAGAGTGTGAGTCGAGCATAGAGATATATA",
         "counts": list(rk.count_kmers_in_read("GAGGAGATTTAGAGTGTGAGTCGAGCATAGAGATATATA", hapA, hapB))},
    ]
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)

    # ---- F2: toy CLI runs (gzip mode = the correct one; see SURVEY §4 trap) -----------
    toy = {}
    for reads_name in ("test.ccs.fastq.gz", "test.fa", "test.fastq"):
        wd = os.path.join(tmp, "cli_" + reads_name)
        os.makedirs(wd)
        out = run_cli(ref, wd, [os.path.join(rdata, reads_name), os.path.join(rdata, "hapA.txt"),
                                os.path.join(rdata, "hapB.txt")])
        entry = {"stdout": out, "files": {}, "counts": []}
        for fn in sorted(os.listdir(wd)):
            body = gzip.open(os.path.join(wd, fn), "rb").read()
            entry["files"][fn] = {
                "size": len(body), "sha256": sha(body),
                "names": [r.name for r in rs.readfq(io.StringIO(body.decode()))],
            }
            if len(body) < 4096:
                entry["files"][fn]["text"] = body.decode()
        for r in rs.open_fastx_read(os.path.join(rdata, reads_name)):
            entry["counts"].append([r.name, *rk.count_kmers_in_read(r.seq, hapA, hapB)])
        toy[reads_name] = entry
    # the --no-gzip-output defect (seq.py:129): record what the reference produces
    wd = os.path.join(tmp, "cli_nogzip")
    os.makedirs(wd)
    out = run_cli(ref, wd, [os.path.join(rdata, "test.ccs.fastq.gz"), os.path.join(rdata, "hapA.txt"),
                            os.path.join(rdata, "hapB.txt"), "--no-gzip-output"])
    toy["test.ccs.fastq.gz --no-gzip-output"] = {
        "stdout": out, "files_created": sorted(os.listdir(wd)),
        "hapA.fastq_sha256": sha(open(os.path.join(wd, "hapA.fastq"), "rb").read()),
        "note": "reference opens the hapB handle on the hapA file name (seq.py:129); hapB file is never created",
    }
    json.dump(toy, open(os.path.join(HERE, "toy_cli.json"), "w"), indent=1)

    # ---- F4: seeded differential vectors --------------------------------------------
    diff = []
    for k in (5, 13, 21, 27, 31, 32):
        rng = random.Random(0xC0FFEE + k)
        n_list = 12 if k == 5 else 60
        pool = set()
        while len(pool) < 2 * n_list:
            pool.add(canonical(rand_seq(rng, k), rk.kmer_to_int))
        pool = sorted(pool)
        rng.shuffle(pool)
        la, lb = pool[:n_list], pool[n_list:]
        # spice: a duplicate line, an entry present in both lists, non-canonical entries
        la.append(la[0])
        lb.append(la[1])
        for lst in (la, lb):
            s = rand_seq(rng, k)
            while canonical(s, rk.kmer_to_int) == s:
                s = rand_seq(rng, k)
            lst.append(s)  # non-canonical: dead entry in the reference
        fa, fb = os.path.join(tmp, f"la{k}.txt"), os.path.join(tmp, f"lb{k}.txt")
        open(fa, "w").write("".join(x + "\n" for x in la))
        open(fb, "w").write("".join(x + "\n" for x in lb))
        ha, hb = rk.create_kmer_hash_set(fa), rk.create_kmer_hash_set(fb)
        reads, counts = [], []
        for i in range(150):
            n = rng.choice([0, 1, k - 1, k, k + 1, rng.randrange(0, 400), rng.randrange(0, 400)])
            s = list(rand_seq(rng, n))
            for _ in range(rng.randrange(0, 6)):
                if n >= k:
                    km = rng.choice(la + lb)
                    if rng.random() < 0.5:
                        km = rc(km)
                    p = rng.randrange(0, n - k + 1)
                    s[p:p + k] = km
            s = "".join(s)
            reads.append(s)
            counts.append(list(rk.count_kmers_in_read(s, ha, hb)))
        entry = {"k": k, "list_a": la, "list_b": lb, "reads": reads, "counts": counts,
                 "num_kmers": [rk.get_number_kmers_in_set(ha), rk.get_number_kmers_in_set(hb)]}
        if k in (21, 32):
            # the same vectors through the reference CLI: TSV + bins
            fq = os.path.join(tmp, f"reads{k}.fa")
            with open(fq, "w") as fh:
                for i, s in enumerate(reads):
                    fh.write(f">r{i} some comment\n{s}\n")
            wd = os.path.join(tmp, f"cli_diff{k}")
            os.makedirs(wd)
            entry["cli_stdout"] = run_cli(ref, wd, [fq, fa, fb])
            entry["cli_bins"] = {
                fn: sha(gzip.open(os.path.join(wd, fn), "rb").read()) for fn in sorted(os.listdir(wd))
            }
        diff.append(entry)
    json.dump(diff, open(os.path.join(HERE, "diff_vectors.json"), "w"))

    # ---- F5: edge vectors ------------------------------------------------------------
    edge = []

    def edge_case(name, la, lb, reads, trailing_newline=True):
        fa, fb = os.path.join(tmp, "ea.txt"), os.path.join(tmp, "eb.txt")
        ta, tb = "\n".join(la), "\n".join(lb)
        if trailing_newline:
            ta, tb = ta + "\n", tb + "\n"
        open(fa, "w").write(ta)
        open(fb, "w").write(tb)
        ha, hb = rk.create_kmer_hash_set(fa), rk.create_kmer_hash_set(fb)
        edge.append({
            "name": name, "text_a": ta, "text_b": tb, "reads": reads,
            "counts": [list(rk.count_kmers_in_read(r, ha, hb)) for r in reads],
            "num_kmers": [rk.get_number_kmers_in_set(ha), rk.get_number_kmers_in_set(hb)],
            "k": ha.contents.k,
        })

    rng = random.Random(77)
    k = 21
    ks = sorted({canonical(rand_seq(rng, k), rk.kmer_to_int) for _ in range(12)})
    a4, b4 = ks[:4], ks[4:8]
    edge_case("kmer_in_both_lists_A_wins", a4, b4 + [a4[0]], [a4[0], rc(a4[0]), "ACGT" + a4[0] + "TTGCA"])
    edge_case("read_shorter_than_k", a4, b4, ["", "A", a4[0][:20], a4[0][1:]])
    edge_case("read_of_length_k", a4, b4, [a4[0], b4[0], rc(b4[1]), ks[9]])
    edge_case("duplicate_lines_counted", a4 + [a4[0], a4[0]], b4, [a4[0] + b4[0]])
    edge_case("no_trailing_newline", a4, b4, [a4[3], b4[3], a4[3] + "A" + b4[3]], trailing_newline=False)
    edge_case("overlapping_hits", a4, b4, [a4[0] + a4[0][-5:] + b4[0], (a4[1] + "C") * 3])
    edge_case("poly_A_and_T_keys", ["A" * 21, a4[0], a4[1], a4[2]], b4, ["A" * 30, "T" * 30, "A" * 21 + "C" + "T" * 21])
    pal = "ACGTACGTACGCGTACGTACGT"[:20]  # even-length palindromes exist only for even k
    edge_case("palindrome_k20", [pal, "AAAAAAAAAAAAAAAAAAAC", "AAAAAAAAAAAAAAAAAACC", "AAAAAAAAAAAAAAAAACCC"],
              ["AAAAAAAAAAAAAAAACCCC", "AAAAAAAAAAAAAAACCCCC", "AAAAAAAAAAAAAACCCCCC", "AAAAAAAAAAAAACCCCCCC"],
              [pal, rc(pal), pal + pal])
    json.dump(edge, open(os.path.join(HERE, "edge_vectors.json"), "w"), indent=1)

    # ---- F6: readfq parser quirks + Read.print ----------------------------------------
    texts = {
        "fasta_basic": ">r1\nACGT\n>r2 comment here\nGG\nTT\n",
        "fastq_basic": "@r1\nACGT\n+\n!!!!\n@r2 c\nGGTT\n+r2\n####\n",
        "header_tab_not_split": ">r1\tx y\nACGT\n",
        "multi_line_fastq": "@r1\nACGT\nTTGA\n+\n!!!!\n####\n@r2\nAC\n+\n!!\n",
        "qual_starts_with_at": "@r1\nACGT\n+\n@!!!\n@r2\nAC\n+\n@@\n",
        "truncated_fastq_to_fasta": "@r1\nACGT\n+\n!!\n",
        "missing_final_newline_fasta": ">r1\nACGT\n>r2\nGGCC",
        "missing_final_newline_fastq": "@r1\nACGT\n+\n!!!!",
        "empty_fastq_read": "@r1\n\n+\n\n@r2\nAC\n+\n!!\n",
        "blank_lines_between": ">r1\nAC\n\nGT\n\n>r2\nTT\n",
        "leading_garbage": "garbage\nmore\n>r1\nACGT\n",
        "empty_input": "",
        "fasta_then_fastq": ">r1\nACGT\n@r2\nGG\n+\n!!\n",
        "empty_name": ">\nACGT\n> x\nGG\n",
        "lowercase_and_n": ">r1\nacgtNNAC\n",
    }
    parsed = {}
    for name, text in texts.items():
        recs = [[r.name, r.seq, r.qual] for r in rs.readfq(io.StringIO(text))]
        printed = io.StringIO()
        for r in rs.readfq(io.StringIO(text)):
            r.print(file=printed)
        parsed[name] = {"text": text, "records": recs, "printed": printed.getvalue()}
    # CRLF goes through a real file in universal-newline text mode, as open_fastx_read does
    crlf = os.path.join(tmp, "crlf.fa")
    open(crlf, "wb").write(b">r1 c\r\nACGT\r\nGG\r\n>r2\r\nTT\r\n")
    parsed["crlf_file"] = {
        "bytes_hex": open(crlf, "rb").read().hex(),
        "records": [[r.name, r.seq, r.qual] for r in rs.open_fastx_read(crlf)],
    }
    json.dump(parsed, open(os.path.join(HERE, "readfq_vectors.json"), "w"), indent=1)

    # fuzz corpus for the parser: random small byte strings over the characters that steer
    # readfq (record markers, newlines of all three kinds, spaces), parsed by the reference
    # from real files in text mode, exactly as open_fastx_read does
    frng = random.Random(20240611)
    alphabet = ["@", ">", "+", "\n", "\n", "\n", "\r\n", "\r", " ", "A", "C", "G", "T", "N", "!", "I", "x", "\t"]
    fuzz = []
    for i in range(400):
        n = frng.randrange(0, 60)
        text = "".join(frng.choice(alphabet) for _ in range(n))
        if frng.random() < 0.5:
            text = frng.choice(["@", ">"]) + "r" + str(i) + " c\n" + text
        fp = os.path.join(tmp, "fuzz.fx")
        open(fp, "wb").write(text.encode())
        recs = [[r.name, r.seq, r.qual] for r in rs.open_fastx_read(fp)]
        fuzz.append({"bytes_hex": text.encode().hex(), "records": recs})
    json.dump(fuzz, open(os.path.join(HERE, "readfq_fuzz.json"), "w"))

    # ---- CLI odds and ends -------------------------------------------------------------
    names = ["x.fastq.gz", "x.fastq", "x.fa.gz", "x.fa", "x.fastqz.gz", "reads.gz", "zz.g", "a.b.fq.gz",
             "dir.d/reads", "reads.fasta.gz", "x.fq.g", "noext", "x.gz.gz", ".gz", "r.fg"]
    misc = {
        "ext_rule": [[n, os.path.splitext(n.rstrip(".gz"))[1]] for n in names],
        "float_str": [],
    }
    # float formatting of count*factor exactly as the reference prints it
    for na, nb, ca, cb in [(4, 3, 4, 1), (4, 3, 0, 2), (3, 4, 7, 7), (300000000, 299999999, 31, 3),
                           (7, 3, 1, 1), (1, 1, 0, 0), (10, 3, 123456, 654321), (3, 10, 2147483647, 2147483647),
                           (299999999, 300000000, 33, 33), (6, 7, 7, 6), (49, 7, 1, 7)]:
        mx = max(na, nb)
        fa_, fb_ = 1.0 * mx / na, 1.0 * mx / nb
        sa, sb = ca * fa_, cb * fb_
        bin_ = "A" if sa > sb else ("B" if sb > sa else "U")
        misc["float_str"].append({"num_a": na, "num_b": nb, "count_a": ca, "count_b": cb,
                                  "score_a": str(sa), "score_b": str(sb), "bin": bin_,
                                  "score_a_hex": sa.hex(), "score_b_hex": sb.hex()})
    json.dump(misc, open(os.path.join(HERE, "cli_misc.json"), "w"), indent=1)

    shutil.rmtree(tmp)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
