"""``seqstats`` orchestration on the GPU path (host drop-in for lib/seq_routines.py).

What the reference does with a pool of ``Consumer`` processes and one task per channel
(lib/seq_routines.py:916-942) happens here in one ``QCContext``: loaders pack reads into CSR batches (or, in
albacore mode, rows of sequencing_summary.txt into record columns), the HIP kernels produce the per-read records and
the run-level tables, and this module turns the tables back into what the reference's consumers expect:

* ``ChannelDict`` -- ``{str(channel): (ReadPerChannel, BasesPerChannel, FailedReads, ChrList, MuxProductivity)}`` or
  ``()`` for an inactive channel (lib/seq_routines.py:905, 940-942; albacore shape :862).  ``ChrList`` (the per-read
  tuples) is not materialised: its only uses are the three histograms and the min/max lengths of ``FastStats``
  (:125-137), which come from the GPU tables carried on the dict (``.tables``).
* ``FastStats(ChannelOut, GCTable=None)`` (:97-279) and ``result_plotting(ch_par)`` (:572-782): the data parts are
  reproduced value for value (including the summary-table quirks, see ``faststats_data``); the figures are rebuilt
  with current plotly, keeping the trace names the reference's injected JavaScript keys on.

No arithmetic on sequence bytes happens here; everything numeric comes out of libpypore_qc.so.
"""
from __future__ import annotations

import ast
import calendar
import math
import os
import re
import shutil
import sys
import time
from decimal import Decimal, ROUND_HALF_UP

import numpy as np

from . import fastq_io
from . import qc
from .logging_module import log, verbosity

I64_MAX = (1 << 63) - 1

# run state, set by run() like the reference's module globals (lib/seq_routines.py:1076-1080)
work_dir = os.getcwd()
prefix = "pypore"
out_dir = "PyPore_results"
summary_flag = False
multiread_flag = False
device = 0
n_channels = 512
n_hours = 48
DIST = None           # (torch.distributed, torch, rank, world) of a `--gpus N` run, set by run(); None on one GPU
H5_OPEN = None        # HDF5 opener override (default: h5py.File); lets other bindings or tests supply the reader


def _h5_opener_from_env():
    """PYPORE_H5_OPENER="module:callable": an HDF5 opener other than h5py.File for the Fast5 loaders (another binding, or
    the stand-in of the test suite in the rank processes of a --gpus run, which no monkeypatch reaches)."""
    spec = os.environ.get("PYPORE_H5_OPENER")
    if not spec:
        return None
    import importlib
    mod, _, attr = spec.partition(":")
    return getattr(importlib.import_module(mod), attr)


# ---------------------------------------------------------------------------------------------------------------
# Python-2 number formatting the summary table depends on
# ---------------------------------------------------------------------------------------------------------------

def py2_round(x, ndigits=0):
    """Python 2 ``round``: half away from zero on the exact decimal value, returns float (:117, :138, :154)."""
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return x
    d = Decimal(repr(x))
    r = d.copy_abs().quantize(Decimal(1).scaleb(-ndigits), rounding=ROUND_HALF_UP)
    return float(-r if d < 0 else r)


def py2_str_float(x):
    """Python 2 ``str(float)`` ('%.12g', with '.0' appended to integral values)."""
    s = "%.12g" % float(x)
    return s + ".0" if not any(c in s for c in ".eni") else s


def local_tz_offset():
    """Offset (s) that ``time.localtime(small_epoch)`` applies on this machine: the reference bins
    ``localtime(t - RefStart)`` (lib/fast5_reader.py:118), i.e. hour of day of the difference in the local zone."""
    return calendar.timegm(time.localtime(0))


# ---------------------------------------------------------------------------------------------------------------
# tables -> ChannelDict
# ---------------------------------------------------------------------------------------------------------------

class ChannelDict(dict):
    """The reference's ChannelDict plus the GPU tables it was built from."""
    tables = None
    summary = False       # albacore shape: 5th element is {hour: (reads, bases)} without the mux level
    n_hours = 48


def channel_dict_from_tables(tables, summary=False):
    nch, nmux, nb = tables["cmh_reads"].shape
    cd = ChannelDict()
    cd.tables, cd.summary, cd.n_hours = tables, summary, nb
    cr, cb, cf = tables["chan_reads"], tables["chan_bases"], tables["chan_fail"]
    cmr, cmb = tables["cmh_reads"], tables["cmh_bases"]
    active = np.nonzero(cr)[0]
    for c in range(nch):
        cd[str(c + 1)] = ()
    for c in active:
        if summary:
            hrs = np.nonzero(cmr[c, 0])[0]
            prod = {str(int(h)): (int(cmr[c, 0, h]), int(cmb[c, 0, h])) for h in hrs}
        else:
            prod = {}
            for m in range(nmux):
                hrs = np.nonzero(cmr[c, m])[0]
                prod[str(m + 1)] = [{str(int(h)): (int(cmr[c, m, h]), int(cmb[c, m, h])) for h in hrs}]
        cd[str(int(c) + 1)] = (int(cr[c]), int(cb[c]), int(cf[c]), None, prod)
    return cd


# ---------------------------------------------------------------------------------------------------------------
# FastStats: data part (lib/seq_routines.py:97-144, 154-155, 207, 228-245)
# ---------------------------------------------------------------------------------------------------------------

def faststats_data(ChannelOut, GCTable=None):
    """Everything ``FastStats`` computes before plotting, from the GPU tables.

    Reproduced verbatim, quirks included: ``PassSum`` is the sum of the per-hour read counts, which contains the
    failed reads, so 'READS TOTAL' = FailSum + PassSum counts them twice (:115, :207); the per-hour mean is
    ``str(round(bases / reads))`` with Python-2 rounding (:117); ``MeanReadLength`` averages the per-hour means over
    ALL bins, empty ones included (the filter ``v != [0, 0]`` never removes a 3-element list, :138-139);
    ``SmallestRead`` is the smallest positive value among the per-channel minima and maxima (:132-137).  The true
    counts are returned beside them (``n_reads``, ``n_fail``, ``n50`` ...)."""
    t = ChannelOut.tables
    nb = ChannelOut.n_hours
    hr, hb = t["hour_reads"], t["hour_bases"]
    FailSum = int(t["chan_fail"].sum())
    PassSum = int(hr.sum())
    HoursDict = []
    for h in range(nb):
        r, b = int(hr[h]), int(hb[h])
        HoursDict.append([r, b, py2_str_float(py2_round(float(b) / float(r))) if (r or b) else 0])
    mm = [int(x) for x in np.concatenate([t["chan_max_len"][t["chan_reads"] > 0], t["chan_min_len"][t["chan_reads"] > 0]])]
    pos = [a for a in mm if a > 0]
    SmallestRead = min(pos) if pos else None      # the reference raises ValueError on min([]) here (:135)
    BiggestRead = max(mm) if mm else None
    MeanReadLength = py2_round(float(np.mean([float(v[2]) for v in HoursDict])), 1)
    TotalThroughput = float(sum(float(v[1]) for v in HoursDict))
    denom = FailSum + PassSum
    pie = None if not denom else [py2_round((float(PassSum) / denom) * 100, 2), py2_round((float(FailSum) / denom) * 100, 2)]
    table = [FailSum + PassSum, PassSum, FailSum, TotalThroughput, MeanReadLength, SmallestRead, BiggestRead]

    # histograms (the reference hands raw vectors to plotly.js, :228-245; the bins below are the GPU's)
    tot = t["totals"]
    n_reads = int(tot[0]) - int(tot[7])           # reads with a valid channel
    w = int(t["lin_bin_width"])
    len_hist = dict(x=[w * i for i in range(len(t["len_hist_lin"]))], width=w, counts=[int(v) for v in t["len_hist_lin"]])
    q_hist = dict(x=[6 + i for i in range(len(t["q_hist"]))], counts=[int(v) for v in t["q_hist"]])
    if GCTable is None:
        gcc = [int(v) for v in t["gc_hist"]]
    else:
        gcc = gc_histogram(GCTable)
    gc_hist = dict(x=[5 * i for i in range(20)], counts=gcc)
    return dict(FailSum=FailSum, PassSum=PassSum, HoursDict=HoursDict, table=table, pie=pie,
                SmallestRead=SmallestRead, BiggestRead=BiggestRead, MeanReadLength=MeanReadLength,
                TotalThroughput=TotalThroughput, len_hist=len_hist, q_hist=q_hist, gc_hist=gc_hist,
                n_reads=n_reads, n_fail=int(tot[2]), n_bases=int(tot[1]), n_gc=int(tot[3]), n_n=int(tot[4]),
                qsum=int(tot[5]), n50=int(t["n50"]), max_len=int(t["max_len"]),
                min_pos_len=None if int(t["min_pos_len"]) >= I64_MAX else int(t["min_pos_len"]))


class GCCounts(list):
    """The 20 bin counts themselves (the merged gc_hist table of a multi-GPU run) where FastStats takes a GC table."""


def gc_histogram(gc_values):
    """20 bins of 5 % over the positive values (``GCVec`` keeps > 0, :136; plotly xbins start 0, size 5, :241).
    ``gc_values`` may be percentages or a (gc_count, length) pair of integer arrays (exact binning)."""
    if isinstance(gc_values, GCCounts):
        return [int(x) for x in gc_values]
    if isinstance(gc_values, tuple):
        gc, ln = (np.asarray(a, dtype=np.int64) for a in gc_values)
        sel = gc > 0
        b = np.minimum(19, (20 * gc[sel]) // np.maximum(ln[sel], 1))
    else:
        v = np.asarray(list(gc_values), dtype=np.float64)
        v = v[v > 0]
        b = np.minimum(19, np.floor(v / 5.0).astype(np.int64))
    return [int(x) for x in np.bincount(b, minlength=20)[:20]]


def _percent(counts):
    s = float(sum(counts))
    return [100.0 * c / s if s else 0.0 for c in counts]


CONFIG = {'displaylogo': False,
          'modeBarButtonsToRemove': ['toImage', 'sendDataToCloud', 'zoom2d', 'pan2d', 'select2d', 'lasso2d',
                                     'hoverClosestCartesian', 'toggleHover', 'autoScale2d']}


def _finish_html(html_file, script):
    """HeatTrigger / HeatTrigger2 (:61-95, :282-318): find the plot div id, append the hover/click glue before
    ``</body></html>`` and move the file into the results directory."""
    with open(html_file, 'r') as fh:
        subject = fh.read()
    m = re.search(r'(?<=<div id=")(.*?)"', subject)
    if m is None:
        log.error("Can't find ID in html file!")
        sys.exit(1)
    fix = script % {"FileId": m.group(1)}
    k = subject.rfind('</body>')
    subject = subject[:k] + fix + subject[k:] if k >= 0 else subject + fix
    with open(html_file, 'w') as fh:
        fh.write(subject)
    dest_dir = os.path.join(work_dir, out_dir)
    os.makedirs(dest_dir, exist_ok=True)
    dest = os.path.join(dest_dir, os.path.basename(html_file))
    if os.path.exists(dest):
        os.remove(dest)
    try:
        shutil.move(html_file, dest_dir)
    except shutil.Error:
        log.error("Unable to move %s" % html_file)
    return dest


_JS_SUMMARY = ("<script type = 'text/javascript'>var graph = document.getElementById('%(FileId)s');"
               "graph.on('plotly_hover', function(data){var divname = data.points.map(function(d){return (d.data.name)});"
               "if(divname == 'ReadsPerHour' || divname == 'BasesPerHour' || divname == 'MeanReadLengthPerHour'){"
               "if(data.xvals){Plotly.Fx.hover(graph, {xval:data.xvals[0] }, ['x5y', 'x5y3','x5y5']);}}});</script>")


def FastStats(ChannelOut, GCTable=None, write_html=True):
    """lib/seq_routines.py:97-279.  Returns the data dict; writes ``<label>_sequencing_summary.html``."""
    log.debug('Preparing data for summary...')
    d = faststats_data(ChannelOut, GCTable)
    if not write_html:
        return d
    import plotly
    import plotly.graph_objs as go
    from plotly.subplots import make_subplots
    log.debug('Plotting stats summary...')
    hours = list(range(len(d["HoursDict"])))
    fig = make_subplots(rows=16, cols=2, shared_xaxes=False,
                        specs=[[{'rowspan': 3}, {}], [None, None], [None, None], [{'rowspan': 3}, {}],
                               [None, None], [None, None], [{'rowspan': 3}, {}], [None, None], [None, None],
                               [None, None], [None, None], [{'colspan': 2}, {}], [None, None], [None, None],
                               [{'colspan': 2}, {}], [None, None]], vertical_spacing=0.001, print_grid=False)
    layout = fig.layout
    pie = d["pie"] or [0.0, 0.0]
    traces = [
        go.Scatter(x=hours, y=[v[0] for v in d["HoursDict"]], name='ReadsPerHour', line=dict(color='rgb(50,136,189)'),
                   showlegend=False, hoverinfo='y', xaxis='x5', yaxis='y1'),
        go.Scatter(x=hours, y=[v[1] for v in d["HoursDict"]], name='BasesPerHour', line=dict(color='rgb(253,174,97)'),
                   showlegend=False, hoverinfo='y', xaxis='x5', yaxis='y3'),
        go.Scatter(x=hours, y=[float(v[2]) for v in d["HoursDict"]], name='MeanReadLengthPerHour',
                   line=dict(color='rgb(21,112,76)'), showlegend=False, hoverinfo='y', xaxis='x5', yaxis='y5'),
        go.Pie(values=pie, labels=['Pass', 'Fail'],
               marker={'colors': ['rgba(25, 133, 37, 1.0)', 'rgba(133, 37, 25, 1.0)']},
               domain={'x': [.70, .82], 'y': [.82, 1]}, name='Reads', hoverinfo='label+percent+name', hole=.4,
               textfont={'color': 'rgb(255,255,255)', 'size': 12}, showlegend=True),
        # the reference lets plotly.js bin raw vectors (histnorm='percent'); here the bins were counted on the GPU
        go.Bar(x=[x + d["len_hist"]["width"] / 2.0 for x in d["len_hist"]["x"]], y=_percent(d["len_hist"]["counts"]),
               width=d["len_hist"]["width"], opacity=0.75, showlegend=False, name='% Reads',
               marker=dict(color='rgb(0, 200, 200)', line=dict(width=0.5, color='rgba(255, 255, 255, 0.3)')),
               xaxis='x4', yaxis='y4', hoverinfo='y+x+name'),
        go.Bar(x=[x + 0.5 for x in d["q_hist"]["x"]], y=_percent(d["q_hist"]["counts"]), width=1, opacity=0.75,
               showlegend=False, name='% Reads',
               marker=dict(color='rgb(0, 0, 255)', line=dict(width=0.5, color='rgba(255, 255, 255, 0.3)')),
               xaxis='x6', yaxis='y6', hoverinfo='y+x+name'),
        go.Bar(x=[x + 2.5 for x in d["gc_hist"]["x"]], y=_percent(d["gc_hist"]["counts"]), width=5, opacity=0.75,
               showlegend=False, name='% Reads',
               marker=dict(color='rgba(154, 0, 0, 1.0)', line=dict(width=0.5, color='rgba(255, 255, 255, 0.3)')),
               xaxis='x7', yaxis='y7', hoverinfo='y+x+name'),
        go.Table(columnorder=[1, 2, 3, 4, 5, 6, 7], columnwidth=[5] * 7, domain={'x': [0, 1], 'y': [0, .12]},
                 header=dict(values=[['<b>READS TOTAL <br> NUMBER</b>'], ['<b>PASS</b>'], ['<b>FAIL</b>'],
                                     ['<b>TOTAL BP</b>'], ['<b>MEAN READ <br>LENGTH (bp)</b>'],
                                     ['<b>SHORTEST READ (bp)</b>'], ['<b>LONGEST READ (bp)</b>']],
                             line=dict(color='#7D7F80'), align=['center'] * 7, fill=dict(color='#00083E'),
                             font=dict(color='white', size=12), height=30),
                 cells=dict(values=[[v] for v in d["table"]], line=dict(color='#7D7F80'), align=['center'] * 7,
                            fill=dict(color='#EDEDEE'), font=dict(color='#506784', size=12), height=25)),
    ]
    fig = go.Figure(data=traces, layout=layout)
    fig.update_layout(title=dict(text='Sperimental Summary', font=dict(size=20)),
                      margin=dict(r=80, t=80, b=80, l=80), hovermode='x', spikedistance=-1,
                      xaxis1=dict(visible=False, showspikes=False), xaxis2=dict(visible=False, showspikes=False),
                      xaxis3=dict(visible=False, showspikes=False),
                      xaxis4=dict(title='Length', hoverformat='.2s', fixedrange=True, domain=[.55, 1]),
                      xaxis5=dict(visible=True, title='hours', showspikes=True, spikethickness=1.5,
                                  spikemode='across', spikecolor='grey', spikedash='solid'),
                      xaxis6=dict(visible=True, title='Quality', hoverformat='.2s', range=[5, 15], fixedrange=True,
                                  domain=[.55, 1]),
                      xaxis7=dict(visible=True, title='GC-Content', hoverformat='.2s', fixedrange=True, domain=[.3, .7]),
                      yaxis1=dict(fixedrange=True, title='Reads', anchor='x5', hoverformat='.2s'),
                      yaxis2=dict(visible=False),
                      yaxis3=dict(title='Bases', fixedrange=True, anchor='x5', hoverformat='.2s'),
                      yaxis4=dict(title='%', hoverformat='.2f', anchor='x4', fixedrange=False, domain=[.632, .732]),
                      yaxis5=dict(title='Mean Length', fixedrange=True, hoverformat='.2s'),
                      yaxis6=dict(title='%', hoverformat='.2f', anchor='x6', fixedrange=True, domain=[.445, .545]),
                      yaxis7=dict(title='%', hoverformat='.2f', anchor='x7', fixedrange=True),
                      legend=dict(x=1, y=1, xanchor='right'))
    out_html_2 = os.path.join(work_dir, prefix + '_sequencing_summary.html')
    plotly.offline.plot(fig, filename=out_html_2, auto_open=False, show_link=False, config=CONFIG)
    d["html"] = _finish_html(out_html_2, _JS_SUMMARY)
    log.info('Finish')
    return d


# ---------------------------------------------------------------------------------------------------------------
# result_plotting: data part (lib/seq_routines.py:572-586, 589-680, 378-394, 507-514, 519-529)
# ---------------------------------------------------------------------------------------------------------------

def nano_layout():
    """The 32 x 16 MinION channel layout of lib/seq_routines.py:589-671, from its arithmetic structure: 8 bands of
    4 rows; in row r of a band the left 8 columns ascend from base + 8r + 1 and the right 8 descend from
    base + 32 + 8r + 8."""
    rows = []
    for base in (0, 448, 384, 320, 256, 192, 128, 64):
        for r in range(4):
            rows.append([str(base + 8 * r + 1 + c) for c in range(8)] + [str(base + 32 + 8 * r + 8 - c) for c in range(8)])
    return rows


def plate_data(ch_par):
    """NanoPlate / NanoBasePlate (:673-680): reads and bases per channel on the layout, 0 for inactive channels."""
    NanoPlate, NanoBasePlate = [], []
    for row in nano_layout():
        line = [ch_par.get(c, ()) for c in row]
        NanoPlate.append([int(x[0]) if len(x) > 0 else 0 for x in line])
        NanoBasePlate.append([int(x[1]) if len(x) > 0 else 0 for x in line])
    return NanoPlate, NanoBasePlate


def mux_fixer(lista):
    """Display helper of :354-375: zeros inside runs of zeros become gaps (None)."""
    n = len(lista)
    new_list = [None] * n
    if lista[0] != 0:
        new_list[0] = lista[0]
    if lista[-1] != 0:
        new_list[-1] = lista[-1]
    half = n // 2
    for x in range(n):
        if 0 < x < half or half - 1 < x < n - 1:
            if lista[x] != 0:
                new_list[x] = lista[x]
            elif lista[x + 1] == 0 and lista[x - 1] != 0:
                new_list[x] = lista[x]
            elif lista[x + 1] != 0 and lista[x - 1] == 0 and not (0 < x < half):
                new_list[x] = lista[x]
    return new_list


def channel_plot_data(c, c_plo):
    """MuxList and the integer fail percentage of one channel (:378-394, 507-514; albacore shape :519-529, 561-564).
    ``FailReads = (fail * 100) / reads`` is Python-2 integer division; 100 for an inactive channel."""
    nb = getattr(c_plo, "n_hours", 48)
    entry = c_plo.get(str(c), ())
    if not getattr(c_plo, "summary", False):
        MuxList = [[0] * nb for _ in range(8)]
        if len(entry) > 0:
            for kk, vv in entry[4].items():
                if vv != [] and 1 <= int(kk) <= 4:
                    l1, l2 = [0] * nb, [0] * nb
                    for u, z in vv[0].items():
                        l1[int(u)], l2[int(u)] = int(z[0]), int(z[1])
                    MuxList[int(kk) - 1], MuxList[int(kk) + 3] = l1, l2
    else:
        MuxList = [[0] * nb, [0] * nb]
        if len(entry) > 0:
            for kk, (nr, nbases) in entry[4].items():
                MuxList[0][int(kk)], MuxList[1][int(kk)] = int(nr), int(nbases)
    FailReads = 100
    if len(entry) > 0:
        FailReads = (int(entry[2]) * 100) // int(entry[0])
    return MuxList, FailReads, 100 - FailReads


_JS_HEAT = ("<script type='text/javascript'>var gd = document.getElementById('%%(FileId)s');"
            "gd.on('plotly_click', function(data){var infotext = data.points.map(function(d){return (d.text)});"
            "var divname = data.points.map(function(d){return (d.data.name)}); "
            "var chan = data.points.map(function(d){return (d.text.split(' ')[1].replace('<br', ''))}); "
            "if(divname == 'ReadNHeat' || divname == 'BaseNHeat'){var per = %(per)d; var total = %(total)d; "
            "var vis = new Array(total).fill(false); for (var i = per*(chan-1); i < per*chan; i++) {vis[i]=true}; "
            "var hv = (gd.data[0].visible === true || gd.data[0].visible === undefined) ? [true, false] : [false, true]; "
            "Plotly.update(gd, {'visible': hv.concat(vis)}, {title: infotext.toString()});};});"
            "gd.on('plotly_hover', function(data){var divname = data.points.map(function(d){return (d.data.name)});"
            "if(%(names)s){if(data.xvals){Plotly.Fx.hover(gd, {xval:data.xvals[0] }, ['x3y2', 'x3y3']);}}});</script>")


def result_plotting(ch_par, write_html=True):
    """lib/seq_routines.py:572-782.  Returns (NanoPlate, NanoBasePlate); writes ``<label>_pore_activity_map.html``
    (the layout table is the MinION one; other channel counts get a near-square grid in channel order)."""
    log.info('Plotting results...')
    nch = len(ch_par)
    if nch == 512:
        layout_rows = nano_layout()
    else:
        ncol = 16 if nch <= 1024 else 50
        layout_rows = [[str(r * ncol + c + 1) for c in range(ncol) if r * ncol + c < nch]
                       for r in range((nch + ncol - 1) // ncol)]
        layout_rows[-1] += ['0'] * (ncol - len(layout_rows[-1]))
    NanoPlate = [[int(ch_par.get(c, ())[0]) if len(ch_par.get(c, ())) > 0 else 0 for c in row] for row in layout_rows]
    NanoBasePlate = [[int(ch_par.get(c, ())[1]) if len(ch_par.get(c, ())) > 0 else 0 for c in row] for row in layout_rows]
    if not write_html:
        return NanoPlate, NanoBasePlate
    import plotly
    import plotly.graph_objs as go
    from plotly.subplots import make_subplots
    colorscale = [[0.0, 'rgb(0,0,0)'], [0.001, 'rgb(140,81,10)'], [0.1111111111111111, 'rgb(140,81,10)'],
                  [0.2222222222222222, 'rgb(191,129,45)'], [0.33333333333, 'rgb(223,194,125)'],
                  [0.44444444444, 'rgb(254,224,139)'], [0.5555555555555555, 'rgb(246,232,195)'],
                  [0.6666666666666666, 'rgb(199,234,229)'], [0.7777777777777778, 'rgb(128,205,193)'],
                  [0.8888888888888888, 'rgb(53,151,143)'], [1.0, 'rgb(1,102,94)']]
    T = lambda m: [list(r) for r in zip(*m)]
    xh, yh, zh = T(layout_rows), T(NanoPlate), T(NanoBasePlate)
    hovertext = [['<b>CHANNEL:</b> {}<br /><b>RN:</b> {}<br /><b>BN:</b> {}'.format(xh[i][j], yh[i][j], zh[i][j])
                  for j in range(len(xh[i]))] for i in range(len(xh))]
    nx, ny = len(layout_rows), len(layout_rows[0])
    heat = dict(x=list(range(1, nx + 1)), y=list(range(1, ny + 1)), text=hovertext, hoverinfo="text", xgap=23 if nx <= 32 else 2,
                ygap=8 if nx <= 32 else 2, colorscale=colorscale, colorbar=dict(y=0.77, len=0.470, exponentformat="SI"))
    fig = make_subplots(rows=13, cols=1, shared_xaxes=False,
                        specs=[[{'rowspan': 6}], [None], [None], [None], [None], [None], [None], [{'rowspan': 2}],
                               [None], [{'rowspan': 2}], [None], [None], [{'rowspan': 1}]],
                        vertical_spacing=0.001, print_grid=False)
    fig.add_trace(go.Heatmap(name="ReadNHeat", z=yh, **heat), 1, 1)
    fig.add_trace(go.Heatmap(name="BaseNHeat", z=zh, visible=False, **heat), 1, 1)
    summary = getattr(ch_par, "summary", False)
    nb = getattr(ch_par, "n_hours", 48)
    hour_vec = list(range(nb))
    colors = ['rgb(50,136,189)', 'rgb(253,174,97)', 'rgb(21,112,76)', 'rgb(165,15,21)']
    per = 4 if summary else 10
    # 10 (4) traces per channel, 5120 at 512 channels: as plain dicts, written without per-trace validation (building
    # them as graph objects took 2.9 of the 3.5 s the whole command needs for 2 Gbases of FASTQ)
    extra = []
    for c in range(1, nch + 1):
        MuxList, FailReads, PassReads = channel_plot_data(c, ch_par)
        half = len(MuxList) // 2
        for k in range(half):
            name = 'Channel' if summary else 'Pore%d' % (k + 1)
            y = mux_fixer(MuxList[k])
            extra.append(dict(type='scatter', y=y, name=name, line=dict(color=colors[k]), legendgroup='Pore%dChr%d' % (k + 1, c),
                              showlegend=False, text=['%s<br>%dh' % (a, b) for a, b in zip(y, hour_vec)],
                              hoverinfo="text+name", fill='tozeroy', visible=False, xaxis='x3', yaxis='y2'))
        for k in range(half):
            name = 'Channel' if summary else 'Pore%d' % (k + 1)
            extra.append(dict(type='scatter', x=hour_vec, y=mux_fixer(MuxList[half + k]), name=name, line=dict(color=colors[k]),
                              legendgroup='Pore%dChr%d' % (k + 1, c), visible=False,
                              text=['%dh' % h for h in hour_vec], hoverinfo="y+text+name", fill='tozeroy',
                              xaxis='x3', yaxis='y3'))
        for val, nm, col in ((FailReads, 'Fail', 'rgba(133, 37, 25, 1.0)'), (PassReads, 'Pass', 'rgba(25, 133, 37, 1.0)')):
            extra.append(dict(type='bar', x=[val], y=['Fail/Pass'], width=[0.4], name=nm, text='%d%% %s' % (val, nm),
                              orientation='h', showlegend=False, visible=False, hoverinfo="text",
                              marker=dict(color=col), xaxis='x4', yaxis='y4'))
    vis = [False] * len(extra)
    fig.update_layout(title='Channel Productivity', spikedistance=-1, hovermode='x', barmode='stack',
                      yaxis1=dict(range=[ny + 0.5, 0.5], visible=False, fixedrange=True),
                      xaxis1=dict(visible=False, fixedrange=True, range=[0.5, nx + 0.5], hoverformat='.2s'),
                      yaxis2=dict(fixedrange=False, title='Reads', hoverformat='.2s', anchor='x3'),
                      xaxis2=dict(visible=False, fixedrange=True),
                      xaxis3=dict(title='hours', range=[0, nb], showspikes=True, spikethickness=1.5,
                                  spikemode='across', spikecolor='grey', spikedash='solid'),
                      yaxis3=dict(fixedrange=False, title='Base Pairs', hoverformat='.2s'),
                      xaxis4=dict(visible=False, fixedrange=True, range=[0, 100]),
                      yaxis4=dict(fixedrange=True, visible=False),
                      legend=dict(x=-0.062, y=0.439, xanchor='right'),
                      updatemenus=[dict(type="buttons", active=-1, x=-0.03, y=0.99, buttons=[
                          dict(label='Read Number', method='update', args=[{'visible': [True, False] + vis}]),
                          dict(label='Base Pair Number', method='update', args=[{'visible': [False, True] + vis}])])])
    out_html = os.path.join(work_dir, prefix + '_pore_activity_map.html')
    fig_dict = fig.to_dict()
    fig_dict["data"] = list(fig_dict["data"]) + extra
    try:
        plotly.offline.plot(fig_dict, filename=out_html, show_link=False, auto_open=False, config=CONFIG, validate=False)
    except Exception as e:   # :775-781
        log.error("Something went wrong during plotting...\n%s" % e)
        sys.exit(1)
    names = "divname == 'Channel'" if summary else \
        "divname == 'Pore1' || divname == 'Pore2' || divname == 'Pore3' || divname == 'Pore4'"
    _finish_html(out_html, (_JS_HEAT % dict(per=per, total=len(extra), names=names)))
    return NanoPlate, NanoBasePlate


# ---------------------------------------------------------------------------------------------------------------
# albacore mode: sequencing_summary.txt (+ FASTQ folder for GC)   lib/seq_routines.py:831-863, 1032-1072
# ---------------------------------------------------------------------------------------------------------------

def summary_reader(filein):
    """Columns of sequencing_summary.txt the reference uses (:836-842, 852): whitespace-split rows, header skipped;
    channel = field 3, start time = field 4; of the slice ``[3:-5]`` element 3 (field 6) is summed as bases and the
    last element (field -6) is compared with 7.00 as the mean quality.  ``ref_start`` starts at 1e9 (:834)."""
    ch, st, ln, mq = [], [], [], []
    with open(filein) as fp:
        fp.readline()
        for line in fp:
            f = line.split()
            if not f:
                continue
            ch.append(int(f[3]))
            st.append(float(f[4]))
            sl = f[3:-5]
            ln.append(int(sl[3]))
            mq.append(float(sl[-1]))
    start = np.asarray(st, dtype=np.float64)
    ref_start = float(min(1000000000.0, start.min())) if len(start) else 1000000000.0
    return dict(channel=np.asarray(ch, dtype=np.int64), start=start, length=np.asarray(ln, dtype=np.int64),
                mean_q=np.asarray(mq, dtype=np.float64), ref_start=ref_start)


def dist_rank_world():
    return (DIST[2], DIST[3]) if DIST is not None else (0, 1)


def is_writer():
    """Rank 0 writes the outputs of a multi-GPU run; every rank holds the merged tables."""
    return dist_rank_world()[0] == 0


def my_share(files):
    """This rank's contiguous share of the job's files (merge.shard_files; everything on one GPU)."""
    rank, world = dist_rank_world()
    files = list(files)
    if world == 1:
        return files
    from .merge import shard_files
    return shard_files(files, rank, world)


def finalize_run(ctx):
    """Run-level tables of the job: ``ctx.finalize()`` on one GPU, the RCCL merge of every rank's tables otherwise
    (merge.finalize_distributed: what the reference does by gathering its workers' results over the result queue,
    lib/seq_routines.py:940-942, and summing over channels in FastStats, :103-108)."""
    if DIST is None:
        return ctx.finalize()
    from .merge import finalize_distributed
    dist, torch = DIST[0], DIST[1]
    return finalize_distributed(ctx, dist, torch)


def job_min(value):
    """MIN of a host integer over the ranks (the same value on one GPU)."""
    if DIST is None:
        return int(value)
    from .merge import allreduce_min_int
    return allreduce_min_int(value, DIST[0], DIST[1])


def summary_channel_dict(rows, dev=None):
    """ChannelDict of albacore mode on the GPU: ``localtime(float(x) - ref_start)`` floors the fp64 difference to
    whole seconds (:846), so exactly that integer is handed to the kernels; the hour-of-day + wrap quirk
    (:847-851) is ``time_mode 1``; there is no mux level (``n_mux = 1``).  With several GPUs every rank takes a
    contiguous share of the rows (``ref_start`` is the minimum over all of them)."""
    sec = np.floor(rows["start"] - rows["ref_start"]).astype(np.int64)
    if np.any(rows["channel"] < 1) or np.any(rows["channel"] > n_channels):
        bad = rows["channel"][(rows["channel"] < 1) | (rows["channel"] > n_channels)][0]
        raise IndexError("channel %d outside 1..%d (the reference raises IndexError at lib/seq_routines.py:842)"
                         % (bad, n_channels))
    if np.any(rows["length"] < 0) or np.any(rows["length"] >= (1 << 32)):
        raise ValueError("sequence length outside 0..2^32-1")
    rank, world = dist_rank_world()
    a, e = rank * len(sec) // world, (rank + 1) * len(sec) // world
    with qc.QCContext(device=device if dev is None else dev, n_channels=n_channels, n_mux=1, n_time_bins=n_hours,
                      bin_seconds=3600, time_mode=1, tz_offset_s=local_tz_offset(), ref_time_s=0, fail_q=7.0,
                      max_reads_hint=e - a) as ctx:
        if e > a:
            ctx.submit_records(rows["length"][a:e].astype(np.uint32), rows["mean_q"][a:e], rows["channel"][a:e].astype(np.uint32),
                               np.ones(e - a, np.uint8), sec[a:e])
        tables = finalize_run(ctx)
    if int(tables["totals"][8]):
        raise KeyError("a read falls in hour bin >= %d (the reference raises KeyError at lib/seq_routines.py:113)" % n_hours)
    return channel_dict_from_tables(tables, summary=True)


MAX_LOADERS = 8     # loader threads that read files into pinned slots (-n asks, this caps: each holds a file's text pinned)


def _submit_fastq_files(ctx, files, host_meta, keep=None, threads=1):
    """Every file through the GPU tokenizer (ppqc_submit_fastq): ``threads`` loader threads (the CLI's -n, at most
    MAX_LOADERS) read files straight into pinned host slots while earlier ones are copied and reduced (staging.Stage:
    the files are submitted in their order whoever read them), the text goes to the card as it is.  A text
    the strict tokenizer refuses (blank lines between records, which SeqIO tolerates) goes through the host packer
    instead, with ``host_meta(packed, path)`` supplying (channel, mux, time).  ``keep``: a list that receives
    (path, packed) of every file, host-packed, for -fq (then every file takes the host packer)."""
    from ._lib import PPQCError
    from .staging import Stage

    def fill(path, slot):
        return path, fastq_io.read_text_into(path, slot)

    def submit(item, d):
        path, text = item
        if keep is None:
            try:
                ctx.submit_fastq(text, slot=d, mux=1)
                return True
            except PPQCError as e:
                if "multiple of 4" not in str(e):
                    raise
                log.debug("%s: %s; using the host packer" % (path, e))
        p = fastq_io.pack_fastq(text)          # copies out of the pinned text into arrays of its own
        if p["n_bad_qual"]:
            raise ValueError("%s: %d quality characters outside '!'..'~' (SeqIO raises ValueError)" % (path, p["n_bad_qual"]))
        if keep is not None:
            p["text"] = p["text"].copy()       # the titles are cut out of it after the slot has been refilled
            keep.append((path, p))
        if not p["n_reads"]:
            return False
        ch, mux, t = host_meta(p, path)
        ctx.submit(p["bases"], p["quals"], p["offsets"], ch, mux, t, None, slot=d)
        ctx.slot_wait(d)                       # these arrays are pageable and die with `p`
        return False

    with Stage(ctx, threads=min(max(1, int(threads or 1)), MAX_LOADERS)) as stage:
        stage.run(files, fill, submit)
    e = ctx.input_errors()
    if e["bad_quality_bytes"]:
        raise ValueError("%d quality characters outside '!'..'~' (SeqIO raises ValueError)" % e["bad_quality_bytes"])


def fastq_gc(files, dev=None, threads=1):
    """``fastq_reader`` over a list of FASTQ files (:40-53): per-record (gc_count, length) from the GPU, or -- several
    GPUs, every rank holding a share of the files -- the merged 20-bin GC histogram (same bins, counted on the device)."""
    with qc.QCContext(device=device if dev is None else dev, n_channels=n_channels, n_mux=1, n_time_bins=n_hours) as ctx:
        _submit_fastq_files(ctx, my_share(files), lambda p, path: (p["channel"], np.ones(p["n_reads"], np.uint8),
                                                                   np.zeros(p["n_reads"], np.int64)),
                            threads=threads)
        if DIST is not None:
            extra = finalize_run(ctx)
            gc_table = GCCounts(int(x) for x in extra["gc_hist"])
        else:
            s = ctx.read_stats()
            extra = ctx.finalize() if ctx.n_reads else None
            gc_table = (s["gc"], s["length"])
    # the reference int()s header field 3 minus its 'ch=' prefix (:51) and dies on anything else
    if extra is not None and int(extra["totals"][7]):
        raise ValueError("%d FASTQ records without a usable ch= header field" % int(extra["totals"][7]))
    return gc_table, extra


def summary_module(fqfolder, summary_table_file, size):
    """lib/seq_routines.py:1032-1072: ChannelDict from the table, GC list from the FASTQ files under pass/ fail/."""
    files = list(fastq_io.fastq_files(fqfolder))
    rows = summary_reader(summary_table_file)
    ChannelOut = summary_channel_dict(rows)
    gc_table, seq_tables = fastq_gc(files, threads=size)
    ChannelOut.seq_tables = seq_tables     # bonus: N50 / N count / quality from the sequences themselves
    if not is_writer():
        return ChannelOut, None, None
    plates = result_plotting(ChannelOut)
    stats = FastStats(ChannelOut, gc_table)
    return ChannelOut, stats, plates


# ---------------------------------------------------------------------------------------------------------------
# FASTQ-only mode (extension): channel and start time from the albacore FASTQ headers
# ---------------------------------------------------------------------------------------------------------------

def fastq_module(fqfolder, Fast_flag, size):
    """FASTQ files whose albacore headers carry ch= and start_time= (no sequencing_summary.txt, no Fast5): RefTime is
    the earliest start time of the job.  Without -fq the text is tokenised on the GPU; with -fq the host packer keeps the
    records for the output file."""
    files = list(fastq_io.fastq_files_any(fqfolder))
    if not files:
        log.error("no FASTQ file under %s" % fqfolder)
        sys.exit(1)
    packed = [] if Fast_flag else None
    t_start = time.perf_counter()
    with qc.QCContext(device=device, n_channels=n_channels, n_mux=4, n_time_bins=n_hours, bin_seconds=3600,
                      time_mode=1, tz_offset_s=0, ref_time_s=0, fail_q=7.0) as ctx:
        def meta(p, path):
            return p["channel"], np.ones(p["n_reads"], np.uint8), p["start_time"]   # no mux in FASTQ headers: mux 1
        _submit_fastq_files(ctx, my_share(files), meta, keep=packed, threads=size)
        t_submitted = time.perf_counter()
        per_read = ctx.read_stats() if Fast_flag else None
        ctx.finalize_extremes()
        tmin, tmax = ctx.time_range()            # over the reads that have a start time
        tmin = job_min(tmin)
        ctx.set_ref_time(tmin if tmin != I64_MAX else 0)
        if DIST is None:
            ctx.finalize_tables()
            ctx.finalize_n50()
            tables = ctx.fetch_tables()
        else:
            tables = finalize_run(ctx)
    log.debug("%d reads, %d bases: files read and submitted in %.3f s, tables %.3f s later"
              % (int(tables["totals"][0]), int(tables["totals"][1]), t_submitted - t_start, time.perf_counter() - t_submitted))
    if int(tables["totals"][7]):
        raise ValueError("%d FASTQ records without a usable ch= header field" % int(tables["totals"][7]))
    if int(tables["totals"][11]):
        raise ValueError("%d FASTQ records without a start_time= header field" % int(tables["totals"][11]))
    ChannelOut = channel_dict_from_tables(tables)
    if Fast_flag:
        from . import fastq_out
        _write_fastq(lambda path: fastq_out.write_passing(packed, per_read, path, n_channels=n_channels, part=DIST is not None))
    if not is_writer():
        return ChannelOut, None, None
    plates = result_plotting(ChannelOut)
    stats = FastStats(ChannelOut)
    return ChannelOut, stats, plates


def _write_fastq(write_part):
    """``<label>.fastq.gz``: written at once on one GPU; with several, every rank writes its passing reads as a part
    whose BGZF blocks end with each channel, and rank 0 puts the parts together channel by channel, ranks in order --
    the reference's own scheme (one BGZF file per channel, concatenated in channel order, lib/seq_routines.py:798-816)."""
    from . import fastq_out
    final = os.path.join(work_dir, out_dir, prefix + '.fastq.gz')
    if DIST is None:
        write_part(final)
        return
    dist, _torch, rank, world = DIST
    write_part(final + '.part%d' % rank)
    dist.barrier()
    if rank == 0:
        fastq_out.stitch_parts([final + '.part%d' % r for r in range(world)], final, n_channels)
    dist.barrier()


# ---------------------------------------------------------------------------------------------------------------
# Fast5 modes: h5py loaders on the CPU pack CSR batches (lib/seq_routines.py:903-1029)
# ---------------------------------------------------------------------------------------------------------------

def albacore_cross_check(alb_len, alb_q, length, flags):
    """Where albacore's own summary attributes and the record disagree.  The reference takes ``sequence_length`` and
    ``mean_qscore`` on trust (lib/fast5_reader.py:55-56, lib/multi_read_fast5_reader.py:104-105) and fails a read on
    ``float(mean_qscore) < 7.0`` (:136); this path recomputes length and mean quality from the FASTQ record.  Counts,
    over the reads whose attributes the loaders found: reads whose ``sequence_length`` is not the number of bases, and
    reads whose fail decision differs.  A cross-check, not an input: the tables are the record's."""
    alb_len, alb_q = np.asarray(alb_len, np.int64), np.asarray(alb_q, np.float64)
    has_len, has_q = alb_len >= 0, ~np.isnan(alb_q)
    out = dict(reads_with_attributes=int((has_len | has_q).sum()),
               length_mismatch=int((has_len & (alb_len != np.asarray(length, np.int64))).sum()),
               fail_mismatch=int((has_q & ((alb_q < 7.0) != ((np.asarray(flags) & 1) != 0))).sum()))
    if out["reads_with_attributes"]:
        log.debug("albacore summary attributes on %(reads_with_attributes)d reads: sequence_length differs from the record "
                  "on %(length_mismatch)d, the fail decision (mean_qscore < 7.0) on %(fail_mismatch)d" % out)
        if out["length_mismatch"] or out["fail_mismatch"]:
            log.warning("albacore's sequence_length / mean_qscore disagree with the FASTQ records on %d / %d reads; "
                        "the tables follow the records" % (out["length_mismatch"], out["fail_mismatch"]))
    return out


ALBACORE_CHECK = None    # the last Fast5 run's cross-check (also in the returned stats as stats["albacore_check"])


def _tables_from_loader(record_batches, ref_time, Fast_flag, compat_flags=0):
    """``record_batches``: iterable of lists of loader records.  A loader thread packs each list straight into a pinned
    host slot (staging.pack_records) while the previous batch is copied and reduced."""
    global ALBACORE_CHECK
    from .staging import Stage, pack_records
    kept, alb = [], []
    with qc.QCContext(device=device, n_channels=n_channels, n_mux=4, n_time_bins=n_hours, bin_seconds=3600,
                      time_mode=1, tz_offset_s=local_tz_offset(), ref_time_s=int(ref_time), fail_q=7.0,
                      compat_flags=compat_flags) as ctx:
        def fill(recs, slot):
            return pack_records(recs, slot) if recs else None

        def submit(b, d):
            ctx.submit(b["bases"], b["quals"], b["offsets"], b["channel"], b["mux"], b["start_time"], b["fail_flag"], slot=d)
            alb.append((b["albacore_len"], b["albacore_q"]))
            if Fast_flag:   # -fq keeps the reads: out of the pinned slot, which is about to be refilled
                kept.append({k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in b.items()})
            return True

        with Stage(ctx) as stage:
            stage.run(record_batches, fill, submit)
        ctx.sync()
        per_read = ctx.read_stats() if Fast_flag else None
        ALBACORE_CHECK = None
        mine = dict(reads_with_attributes=0, length_mismatch=0, fail_mismatch=0)
        if alb and any((a >= 0).any() or (~np.isnan(q)).any() for a, q in alb):
            pr = per_read if per_read is not None else ctx.read_stats(columns=("length", "flags"))
            mine = albacore_cross_check(np.concatenate([a for a, _ in alb]), np.concatenate([q for _, q in alb]),
                                        pr["length"], pr["flags"])
        if DIST is not None:      # every rank has checked its own share: the job's counts are their sums
            from .merge import allreduce_sum_ints
            keys = sorted(mine)
            mine = dict(zip(keys, allreduce_sum_ints([mine[k] for k in keys], DIST[0], DIST[1])))
        if mine["reads_with_attributes"]:
            ALBACORE_CHECK = mine
        tables = finalize_run(ctx)
    return tables, per_read, kept


def fast5module(work_dir_, file_folder, prefix_, tmp_dir, Fast_flag, size, h5_open=None):
    """Single-read Fast5 (lib/seq_routines.py:903-948): files are bucketed by the channel in their name (:909-911),
    RefTime is the smallest exp_start_time of the first file of each leaf directory (:819-828, 882-893)."""
    from . import fast5_reader as f5r
    h5_open = h5_open or H5_OPEN or _h5_opener_from_env()
    files, ref_time = f5r.dir_surfer(file_folder, h5_open=h5_open)
    if ref_time is None:
        log.error("no Fast5 file under %s" % file_folder)
        sys.exit(1)
    # the path check of a channel is made on ONE file of the channel in the whole job, whichever rank reads it
    batches = f5r.iter_record_batches(my_share(files), n_channels, threads=size, h5_open=h5_open, job_files=files)
    tables, per_read, kept = _tables_from_loader(batches, ref_time, Fast_flag)
    if is_writer():
        for c in np.nonzero(tables["chan_reads"] == 0)[0]:
            log.warning('Channel %s inactive!' % int(c))
    ChannelOut = channel_dict_from_tables(tables)
    if Fast_flag:
        from . import fastq_out
        _write_fastq(lambda path: fastq_out.write_passing_batches(kept, per_read, path, n_channels=n_channels, part=DIST is not None))
    if not is_writer():
        return ChannelOut, None, None
    plates = result_plotting(ChannelOut)
    stats = FastStats(ChannelOut)
    stats["albacore_check"] = ALBACORE_CHECK
    return ChannelOut, stats, plates


def multifast5reader(work_dir_, file_folder, prefix_, tmp_dir, Fast_flag, size, h5_open=None):
    """Multi-read Fast5 (lib/seq_routines.py:952-1029): one batch per file; RefStart from the first read's
    tracking_id/exp_start_time (lib/multi_read_fast5_reader.py:137).  Every file is counted (the reference drops one
    file's reads through ``for r in range(en)``, :985, and never fills read_lists for -fq; neither defect is kept)."""
    from . import multi_read_fast5_reader as mf5r
    h5_open = h5_open or H5_OPEN or _h5_opener_from_env()
    files = mf5r.dir_surfer(file_folder)
    ref_start = mf5r.ref_start_of(files, h5_open)        # every rank reads it from the job's first file
    batches = mf5r.iter_record_batches(my_share(files), threads=size, h5_open=h5_open)
    # channel_parser takes the read of a single-read channel under every mux (lib/multi_read_fast5_reader.py:64-67):
    # reproduced, so that the tables are the reference's (PPQC_COMPAT_MULTI_READ_SINGLE; fast5module has no such branch).
    # The rule looks at a channel's read count in the whole job: with several GPUs the ranks add their per-channel
    # counts up first (merge._merge_compat_counts), so `--gpus N` gives the tables of one GPU.
    from ._lib import COMPAT_MULTI_READ_SINGLE
    compat = COMPAT_MULTI_READ_SINGLE
    tables, per_read, kept = _tables_from_loader(batches, ref_start or 0, Fast_flag, compat)
    if is_writer():
        for c in np.nonzero(tables["chan_reads"] == 0)[0]:
            log.warning('Channel %s inactive!' % int(c))
    ChannelOut = channel_dict_from_tables(tables)
    if Fast_flag:
        from . import fastq_out
        _write_fastq(lambda path: fastq_out.write_passing_batches(kept, per_read, path, n_channels=n_channels, part=DIST is not None))
    if not is_writer():
        return ChannelOut, None, None
    plates = result_plotting(ChannelOut)
    stats = FastStats(ChannelOut)
    stats["albacore_check"] = ALBACORE_CHECK
    return ChannelOut, stats, plates


def _lit(s, default=None):
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        return default


def init_distributed(gpus, dev):
    """One process per GPU (started by seqstats.launch_ranks / torchrun: RANK, LOCAL_RANK, WORLD_SIZE in the
    environment), torch.distributed over RCCL.  Returns the device this rank computes on.  PPQC_DIST_BACKEND=gloo keeps
    every rank on the device given with -g: the two-ranks-on-one-GPU rehearsal of the test suite (RCCL refuses two
    ranks on one device)."""
    global DIST
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != gpus:
        log.error("--gpus %d but WORLD_SIZE is %d" % (gpus, world))
        sys.exit(2)
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    backend = os.environ.get("PPQC_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl":
        if n_dev < world:
            log.error("--gpus %d but this node shows %d GPU(s)" % (gpus, n_dev))
            sys.exit(3)
        dev = local
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    DIST = (dist, torch, rank, world)
    return dev


def run(arguments):
    """lib/seq_routines.py:1075-1111: the positional list of strings built by seqstats.ArgsReader
    ``[work_dir, file_folder, prefix, out_dir, ver, Fast_flag, size, summary_flag, albacore_file, multiread_flag]``
    (booleans and the -a list as their ``str()``), plus the optional GPU-path extras
    ``[device, input_format, channels, hours, gpus]``."""
    global summary_flag, work_dir, prefix, out_dir, multiread_flag, device, n_channels, n_hours, DIST
    if not len(arguments) > 4:
        sys.exit(1)
    work_dir, file_folder, prefix, out_dir, ver = arguments[0], arguments[1], arguments[2], arguments[3], arguments[4]
    tmp_dir = os.path.join(work_dir, '.sts_temp')
    Fast_flag = bool(_lit(arguments[5], False)) if len(arguments) > 5 else False
    size = int(arguments[6]) if len(arguments) > 6 else 1
    summary_flag = bool(_lit(arguments[7], False)) if len(arguments) > 7 else False
    multiread_flag = bool(_lit(arguments[9], False)) if len(arguments) > 9 else False
    device = int(arguments[10]) if len(arguments) > 10 else 0
    input_format = arguments[11] if len(arguments) > 11 else 'fast5'
    n_channels = int(arguments[12]) if len(arguments) > 12 else 512
    n_hours = int(arguments[13]) if len(arguments) > 13 else 48
    gpus = int(arguments[14]) if len(arguments) > 14 else 1
    log.info('Start Main Parser')
    verbosity(_lit(ver, 1) if isinstance(ver, str) else ver)
    os.makedirs(os.path.join(work_dir, out_dir), exist_ok=True)
    DIST = None
    if gpus > 1:
        device = init_distributed(gpus, device)
    try:
        if summary_flag:
            summary_table = _lit(arguments[8])
            summary_table = summary_table[0] if isinstance(summary_table, (list, tuple)) else arguments[8]
            return summary_module(file_folder, summary_table, size)
        if input_format == 'fastq':
            return fastq_module(file_folder, Fast_flag, size)
        if not multiread_flag:
            return fast5module(work_dir, file_folder, prefix, tmp_dir, Fast_flag, size)
        return multifast5reader(work_dir, file_folder, prefix, tmp_dir, Fast_flag, size)
    finally:
        if DIST is not None:
            DIST[0].destroy_process_group()
            DIST = None
