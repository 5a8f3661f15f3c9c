"""Writes tests/golden/test_samples_512.csv: the header and the first 512 rows of the reference's own testSamples.csv (sample
DATA, not code), so that the ingest tests that run on the GPU box -- where /root/reference does not exist -- still see real
rows: empty fields, genre strings, floats with two decimals.  Run here: python tests/golden/make_csv_excerpt.py"""
import os

REF_CSV = "/root/reference/src/main/resources/webroot/sampledata/testSamples.csv"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_samples_512.csv")

if __name__ == "__main__":
    with open(REF_CSV, "rb") as f:
        lines = f.read().split(b"\n")
    with open(OUT, "wb") as f:
        f.write(b"\n".join(lines[:513]) + b"\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
