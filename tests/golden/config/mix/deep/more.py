sched = {'_cover_': True, 'warm': 500, 'kind': 'step'}
net = {'depth': -1, 'gone': -2}
