"""Test / baseline infrastructure only -- see the headers of ophelia_oracle.py and oph_cpu.c."""
