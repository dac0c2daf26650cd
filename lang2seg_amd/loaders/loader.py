"""Loader — the dataset index of lib/loaders/loader.py:72-167: `data.json` (refs / images / anns / sentences / word_to_ix /
cat_to_ix / label_length) plus the `/labels` array of `data.h5` (one zero-padded row of word indices per sentence).

Same attributes and methods as the reference class (`Refs`, `Images`, `Anns`, `Sentences`, `annToRef`, `sentToRef`,
`vocab_size`, `label_length`, `encode_labels`, `decode_labels`, `fetch_label`, `fetch_seq`).  `data.h5` (written by
tools/prepro.py:287-289 with h5py) is read by the dependency-free HDF5 reader loaders/h5lite.py - this image has no h5py - which is
pinned by files h5py itself wrote (tests/golden/h5); a `.npy` export of the labels is accepted as well."""
import json
import os
import random

import numpy as np


def load_labels(path):
    """the /labels dataset of data.h5 as an int array [num_sentences][label_length] (lib/loaders/loader.py:103-104)"""
    if path.endswith('.npy'):
        return np.load(path)
    if not os.path.exists(path) and os.path.exists(path + '.npy'):
        return np.load(path + '.npy')
    from .h5lite import read_dataset
    return np.asarray(read_dataset(path, 'labels'))


class Loader(object):
    def __init__(self, data_json, data_h5=None, verbose=True):
        say = print if verbose else (lambda *a: None)
        say('Loader loading data.json:', data_json)
        self.info = json.load(open(data_json))
        self.word_to_ix = self.info['word_to_ix']
        self.ix_to_word = {ix: wd for wd, ix in self.word_to_ix.items()}
        say('vocab size is', self.vocab_size)
        self.cat_to_ix = self.info['cat_to_ix']
        self.ix_to_cat = {ix: cat for cat, ix in self.cat_to_ix.items()}
        say('object cateogry size is', len(self.ix_to_cat))
        self.images, self.anns = self.info['images'], self.info['anns']
        self.refs, self.sentences = self.info['refs'], self.info['sentences']
        say('we have %s images.' % len(self.images))
        say('we have %s anns.' % len(self.anns))
        say('we have %s refs.' % len(self.refs))
        say('we have %s sentences.' % len(self.sentences))
        say('label_length is', self.label_length)
        self.Refs = {ref['ref_id']: ref for ref in self.refs}
        self.Images = {image['image_id']: image for image in self.images}
        self.Anns = {ann['ann_id']: ann for ann in self.anns}
        self.Sentences = {sent['sent_id']: sent for sent in self.sentences}
        self.annToRef = {ref['ann_id']: ref for ref in self.refs}
        self.sentToRef = {sent_id: ref for ref in self.refs for sent_id in ref['sent_ids']}
        self.data_h5 = None
        if data_h5 is not None:
            say('Loader loading data.h5:', data_h5)
            self.data_h5 = {'labels': load_labels(data_h5)}
            assert self.data_h5['labels'].shape[0] == len(self.sentences), 'label.shape[0] not match sentences'
            assert self.data_h5['labels'].shape[1] == self.label_length, 'label.shape[1] not match label_length'

    @property
    def vocab_size(self):
        return len(self.word_to_ix)

    @property
    def label_length(self):
        return self.info['label_length']

    def encode_labels(self, sent_str_list):
        """list of n sentences -> int32 (n, label_length), zero padded, unknown words -> <UNK>"""
        L = np.zeros((len(sent_str_list), self.label_length), dtype=np.int32)
        for i, sent_str in enumerate(sent_str_list):
            for j, w in enumerate(sent_str.split()):
                if j < self.label_length:
                    L[i, j] = self.word_to_ix[w] if w in self.word_to_ix else self.word_to_ix['<UNK>']
        return L

    def decode_labels(self, labels):
        return [' '.join(self.ix_to_word[int(i)] for i in row.tolist() if i != 0) for row in np.asarray(labels)]

    def fetch_label(self, ref_id, num_sents):
        """int32 (num_sents, label_length) and the picked sent_ids (sampled with replacement when the ref has fewer)"""
        sent_ids = list(self.Refs[ref_id]['sent_ids'])
        if len(sent_ids) < num_sents:
            sent_ids += [random.choice(sent_ids) for _ in range(num_sents - len(sent_ids))]
        else:
            sent_ids = sent_ids[:num_sents]
        seq = np.vstack([self.data_h5['labels'][self.Sentences[s]['h5_id'], :] for s in sent_ids])
        return seq, sent_ids

    def fetch_seq(self, sent_id):
        return self.data_h5['labels'][self.Sentences[sent_id]['h5_id'], :]
