"""The command-line front end (andi's interface, src/andi.c + src/io.c) on top of
the C-ABI.  Option handling and input validation run without a GPU; the runs that
compute distances are GPU tests and mirror the reference's shell tests
(test/test_extra.sh, test/test_join.sh, test/nan.sh, test/low_homo.sh)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rand_dna

CLI = os.path.join(ROOT, "andi_amd", "andi-hip")


def fasta(path, records, width=70):
    with open(path, "w") as f:
        for name, seq in records:
            f.write(">%s some comment\n" % name)
            s = seq.decode()
            for k in range(0, len(s), width):
                f.write(s[k:k + width] + "\n")
    return str(path)


def run(args, stdin=None):
    p = subprocess.run([CLI] + args, input=stdin, capture_output=True, timeout=600)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def test_cli_usage_and_validation(tmp_path):
    assert os.path.exists(CLI), "build it with make -C andi_amd/csrc"
    rc, out, err = run(["--help"])
    assert rc == 0 and out.startswith("Usage: andi-hip [OPTIONS...] FILES...") and "--file-of-filenames" in out
    rc, out, err = run(["--version"])
    assert rc == 0 and "andi-hip" in out
    one = fasta(tmp_path / "one.fa", [("A", b"ACGTACGTAC")])
    rc, out, err = run([one])
    assert rc == 1 and "less than two sequences (1 given)" in err  # src/andi.c:265-270
    rc, out, err = run(["-j"])
    assert rc == 1 and "In join mode at least one filename needs to be supplied." in err  # src/andi.c:234
    empty = fasta(tmp_path / "empty.fa", [("A", b"ACGT"), ("B", b"NNNN")])
    rc, out, err = run([empty])
    assert rc == 1 and "The sequence B is empty." in err and "other than acgtACGT" in err  # src/andi.c:302-304,282
    two = fasta(tmp_path / "two.fa", [("A", b"ACGTACGTAC"), ("B", b"ACGTACGTAA")])
    rc, out, err = run(["-p", "7", "-m", "nope", "-b", "x", "--version"])
    assert "between 0 and 1, exclusive" in err and "Ignoring argument for --model" in err
    assert "Expected a positive number for -b" in err
    notfasta = tmp_path / "x.txt"
    notfasta.write_text("hello\n")
    rc, out, err = run([str(notfasta), two])
    assert "x.txt: File must start with '>'." in err  # libs/pfasta.c:317-319


def test_fasta_acceptance_rules(tmp_path):
    """The reader takes and refuses what the reference's parser does (libs/pfasta.c:304-480, src/io.c:196-233),
    with its messages: CR LF, blank lines, blanks inside sequence lines, '>' inside a sequence word, headers
    without a comment; missing '>', empty name, header at the end of the file, text that is not a sequence."""
    def records(text):  # what a file parses to: the run fails later ("less than two sequences") or goes on to the GPU
        f = tmp_path / "t.fa"
        f.write_bytes(text)
        return run([str(f)])

    # accepted: one record only -> "less than two sequences (1 given)" proves it was read as exactly one
    for text in (b">A\r\nACGT\r\nACGT\r\n",                      # CR LF
                 b">A\n\n\nACGT\n\n  \nAC GT\n",                  # blank lines, blanks inside a line
                 b">A comment > with more\nAC>GT\nACGT",             # '>' inside a line is part of the word; no final newline
                 b">A\n-ACGT*\n"):                                   # lines may start with '-' or '*'
        rc, out, err = records(text)
        assert rc == 1 and "less than two sequences (1 given)" in err, (text, err)
    rc, out, err = records(b">A\nACGT\n>B\tcomment\nGGCC\n>C\nTT\n")
    assert "less than two" not in err  # three records
    # refused, with the parser's own words; records before the defect are kept
    for text, msg, given in ((b"", "File is empty.", 0),
                             (b"\n>A\nACGT\n", "File must start with '>'.", 0),
                             (b">A\nACGT\n>\nACGT\n", "Empty name on line 3.", 1),
                             (b">A\nACGT\n>B", "Unexpected EOF in name on line 3.", 1),
                             (b">A\nACGT\n>B comment", "Unexpected EOF in comment on line 3.", 1),
                             (b">A\nACGT\n>B\n\n", "Empty sequence on line 5.", 1),
                             (b">A\nACGT\n>B\n>C\nACGT\n", "Empty sequence on line 4.", 1),
                             (b">A\nACGT\n12 ACGT\n", "Expected '>' but found '1' on line 3.", 1)):
        rc, out, err = records(text)
        assert rc == 1 and ("t.fa: " + msg) in err, (text, err)
        assert ("less than two sequences (%d given)" % given) in err, (text, err)


@pytest.mark.gpu
def test_cli_matrix_equals_oracle_and_options_agree(tmp_path, orc):
    from andi_amd import synth
    seqs, _ = synth.genome_set(5, 60000, 0.002, 0.08, seed=31)
    names = ["S%d" % k for k in range(5)]
    f = fasta(tmp_path / "five.fa", list(zip(names, seqs)))
    rc, out, err = run([f])
    assert rc == 0, err
    M = orc.dist_matrix(seqs, threads=4)
    lines = out.splitlines()
    assert lines[0] == "5"
    for i in range(5):
        cells = lines[1 + i].split()
        assert cells[0] == names[i]
        for j in range(5):
            want = 0.0 if i == j else orc.estimate(M[i, j].astype(np.uint64) + M[j, i], orc.M_JC)
            assert cells[1 + j] == "%1.4f" % want
    # test/test_extra.sh:16-31: --low-memory, threads, file-of-filenames, stdin give the same output
    assert run(["-l", f])[1] == out and run(["-t", "1", f])[1] == out
    parts = [fasta(tmp_path / ("p%d.fa" % k), [(names[k], seqs[k])]) for k in range(5)]
    assert run(parts)[1] == out
    fof = tmp_path / "fof.txt"
    fof.write_text("\n".join(parts) + "\n")
    assert run(["--file-of-filenames", str(fof)])[1] == out
    assert run(["--file-of-filenames", "-"], stdin=fof.read_bytes())[1] == out
    assert run([], stdin=open(f, "rb").read())[1] == out
    # models, -v coverage block, -vv asymmetry
    rc, raw, _ = run(["-m", "Raw", f])
    assert raw.splitlines()[1].split()[2] == "%1.4f" % orc.estimate(M[0, 1].astype(np.uint64) + M[1, 0], orc.M_RAW)
    rc, ani, _ = run(["-m", "ANI", f])
    assert ani.splitlines()[1].split()[2] == "%1.4f" % orc.estimate(
        orc.dist_matrix(seqs, model=orc.M_ANI, threads=4)[[0, 1], [1, 0]].astype(np.uint64).sum(axis=0), orc.M_ANI)
    rc, v, _ = run(["-v", f])
    assert v.startswith(out) and "\nCoverage:\n" in v
    assert v.split("Coverage:\n")[1].splitlines()[0].split()[1] == "%1.4e" % orc.coverage(M[0, 1])
    rc, vv, _ = run(["-vv", f])
    assert vv.splitlines()[1].split()[2] == "%1.4f" % orc.estimate(M[0, 1], orc.M_JC)
    # bootstrap: -b 3 prints 3 matrices in total (src/andi.c:198)
    rc, b, _ = run(["-b", "3", f])
    assert rc == 0 and b.startswith(out) and b.count("\n5\n") + 1 == 3


@pytest.mark.gpu
def test_cli_join_nan_low_homology(tmp_path, orc):
    from andi_amd import synth
    rng = np.random.default_rng(3)
    base = synth.base_codes(90000, 8)
    a = synth.to_bytes(base)
    b = synth.to_bytes(synth.mutate_codes(base, 0.1, 9, raw=True))
    # test/test_join.sh: contigs in separate records, -j joins them and names the genome after the file
    fa = fasta(tmp_path / "first.genome.fa", [("c1", a[:30000]), ("c2", a[30000:70000]), ("c3", a[70000:])])
    fb = fasta(tmp_path / "second.fasta", [("d1", b[:45000]), ("d2", b[45000:])])
    rc, out, err = run(["-j", "-m", "Raw", fa, fb])
    assert rc == 0, err
    rows = [l.split() for l in out.splitlines()[1:]]
    assert rows[0][0] == "first" and rows[1][0] == "second"
    assert abs(float(rows[0][2]) - 0.1) < 0.03
    ja, jb = a[:30000] + b"!" + a[30000:70000] + b"!" + a[70000:], b[:45000] + b"!" + b[45000:]
    M = orc.dist_matrix([ja, jb], model=orc.M_RAW, threads=2)
    assert rows[0][2] == "%1.4f" % orc.estimate(M[0, 1].astype(np.uint64) + M[1, 0], orc.M_RAW)
    # test/nan.sh: unrelated sequences -> nan + warning + failure exit code
    fn = fasta(tmp_path / "nan.fa", [("x", rand_dna(rng, 10000)), ("y", rand_dna(rng, 10000))])
    rc, out, err = run([fn])
    assert rc == 1 and "nan" in out and "reported as nan" in err
    # test/low_homo.sh: 100 shared nt in 100 kbp -> homology warning
    shared = rand_dna(rng, 100)
    fl = fasta(tmp_path / "low.fa", [("x", shared + rand_dna(rng, 100000)), ("y", shared + rand_dna(rng, 100000))])
    rc, out, err = run([fl])
    assert "homology" in err or "reported as nan" in err
    # short sequences warn and fail softly (src/andi.c:306-316)
    fs = fasta(tmp_path / "short.fa", [("x", a[:500]), ("y", b[:500])])
    rc, out, err = run(["--truncate-names", fs])
    assert rc == 1 and "shorter than a thousand" in err and out.splitlines()[0] == "2"
