"""Console entry ``pypore <command> [options]`` (lib/pypore.py:5-61).  Only ``seqstats`` lives on the GPU path;
``fastqgen`` and ``alignment`` are outside this package's scope and say so."""
from __future__ import annotations

import sys

USAGE = '''
usage: pypore [command] [options]

Commands:
seqstats             analyze the entries in the directory given by user, generating
                     an interactive (HTML) experimental summary and, optionally, a FASTQ file.
fastqgen             (not part of the MI355X build: use the reference implementation)
alignment            (not part of the MI355X build: use the reference implementation)
'''


def run_subtool(mod, ags):
    if mod == 'seqstats':
        from . import seqstats as sq
        return sq.run(ags)
    if mod in ('fastqgen', 'alignment'):
        print("pypore: error: command '%s' is outside the GPU QC path of this build" % mod)
        sys.exit(1)
    print(USAGE)
    print("pypore: error: invalid command: '%s' (choose from 'seqstats', 'fastqgen', 'alignment')" % mod)
    sys.exit(1)


def main(argv=None):
    argv = sys.argv if argv is None else argv
    if not len(argv) > 1:
        print(USAGE)
        sys.exit(1)
    return run_subtool(argv[1], [str(a) for a in argv[2:]])


if __name__ == "__main__":
    main()
