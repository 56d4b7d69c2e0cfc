"""The design notes cite files (profiles/, scripts/, tests/, sources, the reference's file:line): every one must exist
 -- scripts/check_citations.py; and no line of them is longer than 160 characters."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_cited_file_exists():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_citations.py"), "-v"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert int(r.stdout.split()[0]) > 200   # (the scan found the documents)


def test_design_notes_keep_to_160_columns():
    docs = [os.path.join(ROOT, "DESIGN.md")] + glob.glob(os.path.join(ROOT, "docs", "design", "*.md"))
    assert len(docs) >= 8
    for d in docs:
        with open(d) as f:
            fence = False
            for no, line in enumerate(f, 1):
                if line.strip().startswith("```"):
                    fence = not fence
                assert fence or len(line.rstrip("\n")) <= 160, f"{os.path.relpath(d, ROOT)}:{no} has {len(line) - 1} characters"
