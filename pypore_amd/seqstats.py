"""``pypore seqstats`` command line: the options, defaults and error exits of lib/seqstats.py:54-127, marshalled to
``seq_routines.run`` as the same positional list of strings (lib/seqstats.py:50-52).

Options added for the GPU path (they do not change any existing one): ``-g/--gpu_device`` (HIP device index),
``-f/--input_format`` (``fast5`` as in the reference, or ``fastq`` to read FASTQ files whose albacore headers carry
``ch=`` / ``start_time=`` without a sequencing_summary.txt), ``--channels`` and ``--hours`` (table sizes; 512 and 48
are hard-coded in the reference, lib/seq_routines.py:99,833) and ``--gpus N``: N processes, one per GPU, each over a
contiguous share of the input files, their tables merged over RCCL -- what ``-n`` is to the reference's pool of worker
processes (lib/seqstats.py:111-112, lib/seq_routines.py:916-923); ``-n`` itself stays the number of loader threads of
each process."""
from __future__ import annotations

import argparse
import atexit
import os
import shutil
import sys
import tempfile

from .logging_module import log, attach_file_handler

BANNER = '''\r      \n
                                  ________________
                                 |                |
                                 |    #####       |
                                 |    ##  ##      |
                                 |    #####       |
                                 |    ## #####    |
                                 |    ## ##  ##   |
                                 |       #####    |
                                 |       ##       |
                                 |       ##       |
                                 |________________|

                                       PyPore
                                      SeqStats

usage: %(prog)s [-i <input_directory>] [-l <my_label>] [options]'''


class readable_dir(argparse.Action):
    """lib/seqstats.py:10-20: exit(1) unless the value is a readable directory."""

    def __call__(self, parser, namespace, values, option_string=None):
        if not os.path.isdir(values) or not os.access(values, os.R_OK):
            log.error("readable_dir:{0} is not a directory".format(values))
            sys.exit(1)
        setattr(namespace, self.dest, values)


def build_parser(default_dir):
    parser = argparse.ArgumentParser(add_help=False, prog='seqstats', usage=BANNER,
                                     formatter_class=argparse.RawDescriptionHelpFormatter)
    g = parser.add_argument_group(title='mandatory arguments', description='''\
-i,  --input_directory             path to the file folder
-l,  --label                       label for output file
''')
    g.add_argument('-i', '--input_directory', action=readable_dir, default=default_dir, help=argparse.SUPPRESS)
    g.add_argument('-l', '--label', action='store', metavar='', help=argparse.SUPPRESS)
    f = parser.add_argument_group(title='options', description='''\
-fq, --fastq                       generate fastq.gz [no]
-a,  --albacore_summary            if provided, it is used to gather the sequencing
                                   informations (fastq, not fast5, folder required)
-m,  --multi_read_fast5            activate the multi read fast5 reading [no]
-o,  --output_dir                  if omitted, generates a results directory in the
                                   current position
-n,  --threads_number              number of loader threads [1]
-v,  --verbose                     increase verbosity: 0 = warnings only, 1 = info,
                                   2 = debug, 3 = debug on terminal. No number means info
-g,  --gpu_device                  HIP device index [0]
-f,  --input_format                fast5 | fastq (FASTQ files with albacore headers) [fast5]
     --channels                    number of flow-cell channels [512]
     --hours                       number of hourly bins [48]
     --gpus                        number of GPUs, one process each [1]
-h,  --help                        show this help message and exit
''')
    f.add_argument('-fq', '--fastq', action="store", nargs=1, metavar='', choices=['y', 'n', 'yes', 'no'],
                   default=['no'], help=argparse.SUPPRESS)
    f.add_argument('-a', '--albacore_summary', action="store", nargs=1, metavar='', default=None,
                   help=argparse.SUPPRESS)
    f.add_argument('-m', '--multi_read_fast5', nargs=1, metavar='', choices=['y', 'n', 'yes', 'no'], default=['no'],
                   help=argparse.SUPPRESS)
    f.add_argument('-o', '--output_dir', action="store", nargs=1, metavar='', help=argparse.SUPPRESS)
    f.add_argument('-n', '--threads_number', action="store", type=int, default=1, metavar='', help=argparse.SUPPRESS)
    f.add_argument('-h', '--help', action="help", help=argparse.SUPPRESS)
    f.add_argument('-v', '--verbose', const=1, default=1, type=int, nargs="?", help=argparse.SUPPRESS,
                   choices=range(0, 4))
    f.add_argument('-g', '--gpu_device', action="store", type=int, default=0, metavar='', help=argparse.SUPPRESS)
    f.add_argument('-f', '--input_format', action="store", choices=['fast5', 'fastq'], default='fast5', metavar='',
                   help=argparse.SUPPRESS)
    f.add_argument('--channels', action="store", type=int, default=512, metavar='', help=argparse.SUPPRESS)
    f.add_argument('--hours', action="store", type=int, default=48, metavar='', help=argparse.SUPPRESS)
    f.add_argument('--gpus', action="store", type=int, default=1, metavar='', help=argparse.SUPPRESS)
    return parser


def ArgsReader(args):
    """lib/seqstats.py:23-52: flags to booleans, everything to strings, then seq_routines.run."""
    work_dir = os.getcwd()
    fastq_flag = args.fastq[0] in ['y', 'yes']
    out_dir = str(args.output_dir[0]) if args.output_dir is not None else 'PyPore_results'
    albacore_file = args.albacore_summary
    summary_flag = albacore_file is not None
    multi_read_flag = args.multi_read_fast5[0] in ['y', 'yes']
    ags = [work_dir, args.input_directory, args.label, out_dir, str(args.verbose), str(fastq_flag),
           str(args.threads_number), str(summary_flag), str(albacore_file), str(multi_read_flag),
           # extras of the GPU path (absent = defaults, so a reference-shaped 10-element list still works)
           str(args.gpu_device), args.input_format, str(args.channels), str(args.hours), str(args.gpus)]
    from . import seq_routines as sq
    return sq.run(ags)


def launch_ranks(gpus, argsin):
    """``--gpus N`` started plainly: N rank processes as ONE torchrun child (``python -m torch.distributed.run
    --nproc-per-node N -m pypore_amd.pypore seqstats ...``), decided before this process has imported torch or touched
    HIP.  A child process, never an exec of this one; its exit code is ours, so a rank that fails fails the command."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "pypore_amd.pypore", "seqstats"] + list(argsin)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        log.error("the %d-rank run failed (exit code %d)" % (gpus, rc))
        sys.exit(rc if rc > 0 else 1)
    return None


def run(argsin):
    ldir = tempfile.mkdtemp()
    atexit.register(lambda d=ldir: shutil.rmtree(d, ignore_errors=True))
    attach_file_handler()
    parser = build_parser(ldir)
    try:
        args = parser.parse_args(argsin)
        if not args.input_directory or not args.label:
            parser.print_help()
            log.error('Missing mandatory arguments!')
            sys.exit(1)
        if args.gpus < 1:
            log.error('--gpus must be at least 1')
            sys.exit(1)
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            return launch_ranks(args.gpus, argsin)     # nothing here has touched torch or the GPU
        return ArgsReader(args)
    except IOError as msg:
        parser.error(str(msg))
