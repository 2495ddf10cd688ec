"""`Evaluator`: run a list of metrics over every (tracker, sequence, class) of a dataset and print / return / save the summary.
Mirrors how the reference drives its vendored TrackEval (/root/reference/tracker/track.py:196-227, trackeval/eval.py:18-225):
`Evaluator(eval_config).evaluate([dataset], [HOTA(cfg), CLEAR(cfg), Identity(cfg)]) -> (output_res, output_msg)` with
`output_res[dataset][tracker][seq | 'COMBINED_SEQ'][class][metric] = {field: value}`."""
import os
import time

import numpy as np

from .metrics import Count


class Evaluator:
    @staticmethod
    def get_default_eval_config():
        return {'USE_PARALLEL': False, 'NUM_PARALLEL_CORES': 8, 'BREAK_ON_ERROR': True, 'RETURN_ON_ERROR': False, 'LOG_ON_ERROR': None,
                'PRINT_RESULTS': True, 'PRINT_ONLY_COMBINED': False, 'PRINT_CONFIG': True, 'TIME_PROGRESS': True, 'DISPLAY_LESS_PROGRESS': True,
                'OUTPUT_SUMMARY': True, 'OUTPUT_EMPTY_CLASSES': True, 'OUTPUT_DETAILED': True, 'PLOT_CURVES': False}

    def __init__(self, config=None):
        self.config = self.get_default_eval_config()
        self.config.update({k: v for k, v in (config or {}).items() if k in self.config})

    def evaluate(self, dataset_list, metrics_list, show_progressbar=False):
        metrics_list = list(metrics_list) + [Count()]
        names = [m.get_name() for m in metrics_list]
        output_res, output_msg = {}, {}
        for dataset in dataset_list:
            dname = dataset.get_name()
            output_res[dname], output_msg[dname] = {}, {}
            trackers, seqs, classes = dataset.get_eval_info()
            print('\nEvaluating %d tracker(s) on %d sequence(s) for %d class(es) on %s dataset using the following metrics: %s\n'
                  % (len(trackers), len(seqs), len(classes), dname, ', '.join(names)))
            for tracker in trackers:
                t0 = time.time()
                try:
                    res = {}
                    for seq in sorted(seqs):
                        raw = dataset.get_raw_seq_data(tracker, seq)
                        res[seq] = {}
                        for cls in classes:
                            data = dataset.get_preprocessed_seq_data(raw, cls)
                            res[seq][cls] = {m.get_name(): m.eval_sequence(data) for m in metrics_list}
                    res['COMBINED_SEQ'] = {cls: {m.get_name(): m.combine_sequences({s: res[s][cls][m.get_name()] for s in seqs})
                                                 for m in metrics_list} for cls in classes}
                    if self.config['PRINT_RESULTS']:
                        for cls in classes:
                            self._print(tracker, cls, res, seqs, metrics_list)
                    if self.config['OUTPUT_SUMMARY']:
                        for cls in classes:
                            self._write_summary(dataset.get_output_fol(tracker), cls, res['COMBINED_SEQ'][cls], metrics_list)
                    output_res[dname][tracker] = res
                    output_msg[dname][tracker] = 'Success'
                    print('\nAll sequences for %s finished in %.2f seconds' % (tracker, time.time() - t0))
                except Exception as err:
                    output_res[dname][tracker] = None
                    output_msg[dname][tracker] = 'Unknown error occurred.' if not str(err) else str(err)
                    if self.config['BREAK_ON_ERROR']:
                        raise
                    if self.config['RETURN_ON_ERROR']:
                        return output_res, output_msg
        return output_res, output_msg

    @staticmethod
    def _scalar(v):
        """array fields (HOTA over the 19 alphas) are reported as their mean, like the reference's summary"""
        return float(np.mean(v)) if isinstance(v, np.ndarray) else float(v)

    @staticmethod
    def _is_count(metric, field):
        """counts are printed as they are, ratios as percentages"""
        return field in getattr(metric, 'integer_fields', ()) or field in getattr(metric, 'integer_array_fields', ())

    def _print(self, tracker, cls, res, seqs, metrics_list):
        rows = ([] if self.config['PRINT_ONLY_COMBINED'] else sorted(seqs)) + ['COMBINED_SEQ']
        for m in metrics_list:
            fields = m.summary_fields
            print('\n%-34s' % ('%s: %s-%s' % (m.get_name(), tracker, cls)) + ''.join('%-10s' % f for f in fields))
            for s in rows:
                r = res[s][cls][m.get_name()]
                print('%-34s' % ('COMBINED' if s == 'COMBINED_SEQ' else s) +
                      ''.join(('%-10d' % int(self._scalar(r[f])) if self._is_count(m, f) else '%-10.5g' % (100 * self._scalar(r[f]))) for f in fields))

    def _write_summary(self, folder, cls, combined, metrics_list):
        os.makedirs(folder, exist_ok=True)
        fields, vals = [], []
        for m in metrics_list:
            for f in m.summary_fields:
                fields.append(f)
                v = combined[m.get_name()][f]
                vals.append('%d' % int(self._scalar(v)) if self._is_count(m, f) else '%.5g' % (100 * self._scalar(v)))
        with open(os.path.join(folder, cls + '_summary.txt'), 'w') as f:
            f.write(' '.join(fields) + '\n' + ' '.join(vals) + '\n')
